from gym.spaces import MultiDiscrete
