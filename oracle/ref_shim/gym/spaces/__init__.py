import numpy as np
class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self.np_random = np.random.RandomState()
    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed); return [seed]
class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None: shape = np.asarray(low).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
    def sample(self): return self.np_random.uniform(-1,1,self.shape).astype(self.dtype)
    def __eq__(self, o): return isinstance(o, Box) and self.shape==o.shape and np.allclose(self.low,o.low) and np.allclose(self.high,o.high)
class Discrete(Space):
    def __init__(self, n):
        self.n = n; super().__init__((), np.int64)
    def sample(self): return self.np_random.randint(self.n)
    def __eq__(self, o): return isinstance(o, Discrete) and self.n==o.n
class MultiDiscrete(Space):
    def __init__(self, nvec): self.nvec=np.asarray(nvec); super().__init__(self.nvec.shape, np.int64)
class MultiBinary(Space):
    def __init__(self, n): self.n=n; super().__init__((n,), np.int8)
class Dict(Space):
    def __init__(self, spaces=None): self.spaces=spaces; super().__init__(None,None)
class Tuple(Space):
    def __init__(self, spaces=()): self.spaces=spaces; super().__init__(None,None)
class _U:
    @staticmethod
    def flatdim(space):
        if isinstance(space, Box): return int(np.prod(space.shape))
        if isinstance(space, Discrete): return int(space.n)
        raise NotImplementedError
utils = _U()
