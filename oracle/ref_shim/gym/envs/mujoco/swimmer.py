from gym.envs.mujoco.mujoco_env import MujocoEnv
class SwimmerEnv(MujocoEnv): pass
