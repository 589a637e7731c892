# minimal oracle shim (lives in /tmp only) -- just enough surface for stable_baselines3 import
import numpy as np
from gym import spaces
from gym.spaces import Space
class Env:
    metadata = {}
    observation_space = None
    action_space = None
    def seed(self, seed=None): return [seed]
    def close(self): pass
    @property
    def unwrapped(self): return self
class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.metadata = getattr(env, "metadata", {})
    def __getattr__(self, name):
        return getattr(self.env, name)
    def step(self, a): return self.env.step(a)
    def reset(self, **kw): return self.env.reset(**kw)
    def seed(self, seed=None): return self.env.seed(seed)
    def close(self): return self.env.close()
class ObservationWrapper(Wrapper): pass
class RewardWrapper(Wrapper): pass
class ActionWrapper(Wrapper): pass
class GoalEnv(Env): pass
_registry = {}
def make(id, **kw):
    import importlib
    ep = _registry[id]
    mod, cls = ep["entry_point"].split(":")
    env = getattr(importlib.import_module(mod), cls)(**kw)
    import types
    env.spec = types.SimpleNamespace(id=id, max_episode_steps=ep["max_episode_steps"])
    if ep["max_episode_steps"] is not None:
        env = TimeLimit(env, ep["max_episode_steps"])
    return env
class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps):
        super().__init__(env); self._max_episode_steps = max_episode_steps; self._elapsed_steps = None
    def step(self, action):
        o, r, d, info = self.env.step(action); self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info['TimeLimit.truncated'] = not d; d = True
        return o, r, d, info
    def reset(self, **kw):
        self._elapsed_steps = 0; return self.env.reset(**kw)
from gym import envs, utils
