from gym.envs.mujoco.mujoco_env import MujocoEnv
class HalfCheetahEnv(MujocoEnv): pass
