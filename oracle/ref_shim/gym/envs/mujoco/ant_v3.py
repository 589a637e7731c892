from gym.envs.mujoco.mujoco_env import MujocoEnv
class AntEnv(MujocoEnv): pass
