from gym.envs.mujoco import mujoco_env, ant_v3, half_cheetah, swimmer, walker2d
