from gym.spaces import Box
