from gym.spaces import Discrete
