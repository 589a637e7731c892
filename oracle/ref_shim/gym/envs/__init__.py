from gym.envs import registration
