from gym.spaces import Space
