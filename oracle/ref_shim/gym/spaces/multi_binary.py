from gym.spaces import MultiBinary
