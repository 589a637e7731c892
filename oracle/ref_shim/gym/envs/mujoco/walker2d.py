from gym.envs.mujoco.mujoco_env import MujocoEnv
class Walker2dEnv(MujocoEnv): pass
