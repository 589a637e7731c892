"""Minimal gym.spaces stand-ins (gym is not a dependency of the hot path): Box and Discrete with the attributes the
reference reads (shape, low, high, dtype, n; ref: icrl/icrl.py:72-78, on_policy_algorithm.py:381-382)."""
import numpy as np


class Space:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

    def __eq__(self, o):
        return isinstance(o, Box) and self.shape == o.shape and np.allclose(self.low, o.low) and np.allclose(self.high, o.high)

    def __repr__(self):
        return f"Box{self.shape}"


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def __eq__(self, o):
        return isinstance(o, Discrete) and self.n == o.n

    def __repr__(self):
        return f"Discrete({self.n})"
