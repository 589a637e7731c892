"""Vectorised-environment surface of the reference, device-resident.

Mirrors (same names / argument meaning):
  VecEnv, VecEnvWrapper        ref: stable_baselines3/common/vec_env/base_vec_env.py:48-342
  VecCostWrapper               ref: stable_baselines3/common/vec_env/vec_cost_wrapper.py:7-101
  VecNormalize(WithCost)       ref: stable_baselines3/common/vec_env/vec_normalize.py:9-278
  RunningMeanStd               ref: stable_baselines3/common/running_mean_std.py:6-39
  sync_envs_normalization      ref: stable_baselines3/common/vec_env/__init__.py:50-65
  HipSynthVecEnv               stands in for SubprocVecEnv + the MuJoCo envs (subproc_vec_env.py:53-177); spec SURVEY §8d

Observations / rewards / dones are torch tensors in HBM (float64 / float64 / uint8), not numpy arrays: the host only
enqueues kernels.  ``infos`` is a BatchedInfos object that still answers ``infos[i]['cost']`` like the reference's list of
dicts.  Arithmetic lives in libicrl_hip.so (icrl_synth_env_step, icrl_cost_mlp_forward, icrl_vecnorm_step).
"""
import copy
import pickle
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import _lib, spaces
from .structs import EnvT, NormT, p

# kind -> obs, act (width of the action vector handed to the env), max_episode_steps, reward form (icrl_env_t.reward_form)
KINDS = {"hc": (18, 6, 1000, 0), "ant": (113, 8, 500, 1),
         # LapGridWorld / ConstrainedLapGridWorld restated exactly (custom_envs/envs/lap_grid_world.py:29-240): Discrete(2)
         "lgw": (1, 1, 200, 2), "clgw": (1, 1, 200, 3)}
DISCRETE_ACTIONS = {"lgw": 2, "clgw": 2}
ENV_IDS = {  # reference gym ids (custom_envs/__init__.py:43-57,194-224,357-370) -> (kind, early termination, broken)
    "HCWithPos-v0": ("hc", False, False), "HCWithPosTest-v0": ("hc", True, False),
    "AntWall-v0": ("ant", False, False), "AntWallTest-v0": ("ant", True, False),
    "AntWallBroken-v0": ("ant", False, True), "AntWallBrokenTest-v0": ("ant", True, True),
    "LGW-v0": ("lgw", False, False), "CLGW-v0": ("clgw", True, False),
}


def dynamics_matrix(kind):
    """B ~ N(0, 0.05^2) drawn once from RandomState(1234)."""
    o, a, _, _ = KINDS[kind]
    if kind in DISCRETE_ACTIONS:
        return np.zeros((o, a), np.float64)      # no linear dynamics: the grid world is stepped exactly
    return (np.random.RandomState(1234).randn(o, a) * 0.05).astype(np.float64)


class BatchedInfos:
    """infos[i][key] view over per-key device tensors (the reference returns a list of per-env dicts)."""

    def __init__(self, n):
        self.n, self.batch = n, {}

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {k: v[i].item() for k, v in self.batch.items()}


class VecEnv(ABC):
    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def step_async(self, actions): ...

    @abstractmethod
    def step_wait(self): ...

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        pass

    def seed(self, seed=None):
        return [None] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name)] * self.num_envs

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *a, indices=None, **k):
        return [getattr(self, method_name)(*a, **k)]

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self


class VecEnvWrapper(VecEnv):
    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(venv.num_envs, observation_space or venv.observation_space, action_space or venv.action_space)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def reset(self):
        return self.venv.reset()

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def close(self):
        return self.venv.close()

    def __getattr__(self, name):
        if name.startswith("_") or name == "venv":
            raise AttributeError(name)
        return getattr(self.venv, name)


def _as_device_f32(x, device):
    if torch.is_tensor(x):
        return x.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device=device)


class HipSynthVecEnv(VecEnv):
    """N synthetic HCWithPos-/AntWall-shaped envs (or exact LapGridWorlds) stepped by one kernel launch; float64 state in HBM."""

    def __init__(self, n_envs, kind="hc", seed=0, env_index_offset=0, wall_terminate=False, broken=False, device="cuda"):
        self.kind, self.device = kind, torch.device(device)
        o, a, ms, rf = KINDS[kind]
        self.obs_dim, self.act_dim, self.max_steps, self.reward_form = o, a, ms, rf
        self.wall_terminate, self.broken = bool(wall_terminate), bool(broken)
        if kind in DISCRETE_ACTIONS:
            # ref: lap_grid_world.py:50-53 — Box(0, 40) float32 observation space (the env itself emits 2 pos / 40 - 1)
            super().__init__(n_envs, spaces.Box(0.0, 40.0, (o,), np.float32), spaces.Discrete(DISCRETE_ACTIONS[kind]))
        else:
            super().__init__(n_envs, spaces.Box(-np.inf, np.inf, (o,), np.float64), spaces.Box(-1.0, 1.0, (a,), np.float32))
        dev = self.device
        self.B = torch.as_tensor(dynamics_matrix(kind), device=dev)
        self.s = torch.zeros(n_envs, o, dtype=torch.float64, device=dev)
        self.t_ep = torch.zeros(n_envs, dtype=torch.int32, device=dev)
        self.step_count = torch.zeros(n_envs, dtype=torch.int32, device=dev)       # uint32 bits
        self.key = torch.zeros(n_envs, dtype=torch.int32, device=dev)
        self.raw_rew = torch.zeros(n_envs, dtype=torch.float64, device=dev)
        self.dones = torch.zeros(n_envs, dtype=torch.uint8, device=dev)
        self._actions = None
        self._index_offset = int(env_index_offset)
        self.seed(seed)

    @classmethod
    def make(cls, env_id, n_envs, seed=0, device="cuda", env_index_offset=0):
        kind, wall, broken = ENV_IDS[env_id]
        return cls(n_envs, kind, seed, env_index_offset, wall, broken, device)

    def seed(self, seed=None, env_index_offset=None):
        """env i gets stream key seed + (global index of env i) (ref: icrl/utils.py:256-263, subproc_vec_env.py:115-118).  The
        global index is i + the shard offset given at construction (rank * envs per rank on a multi-GPU run), so re-seeding
        through the wrapper chain (PPOLagrangian._setup_model -> env.seed(seed)) keeps every rank on its own shard."""
        seed = 0 if seed is None else int(seed)
        if env_index_offset is not None:
            self._index_offset = int(env_index_offset)
        keys = (np.arange(self.num_envs, dtype=np.int64) + seed + self._index_offset) & 0xFFFFFFFF
        self.key.copy_(torch.as_tensor(keys.astype(np.uint32).view(np.int32), device=self.device))
        self.step_count.zero_(); self.t_ep.zero_(); self.s.zero_()
        return [seed + self._index_offset + i for i in range(self.num_envs)]

    def struct(self):
        return EnvT(self.num_envs, self.obs_dim, self.act_dim, self.max_steps, self.reward_form, int(self.wall_terminate),
                    int(self.broken), 0, p(self.B), p(self.s), p(self.t_ep), p(self.step_count), p(self.key))

    def reset(self):
        e = self.struct()
        _lib.check(_lib.lib().icrl_synth_env_reset(_lib.byref(e), _lib.current_stream()), "icrl_synth_env_reset")
        return self.s.clone()

    def step_async(self, actions):
        self._actions = _as_device_f32(actions, self.device).reshape(self.num_envs, self.act_dim)

    def step_wait(self):
        e = self.struct()
        _lib.check(_lib.lib().icrl_synth_env_step(_lib.byref(e), p(self._actions), p(self.raw_rew), p(self.dones),
                                                  _lib.current_stream()), "icrl_synth_env_step")
        return self.s.clone(), self.raw_rew.clone(), self.dones.clone(), BatchedInfos(self.num_envs)


class VecCostWrapper(VecEnvWrapper):
    """cost = cost_function(previous raw obs, current action) written to infos['cost'] (ref: vec_cost_wrapper.py:51-77)."""

    def __init__(self, venv):
        super().__init__(venv)
        self.cost_function = None
        self.previous_obs = None
        self.actions = None

    def set_cost_function(self, cost_function):
        self.cost_function = cost_function

    def constraint_net(self):
        """the device ConstraintNet behind cost_function, if that is what it is (enables the fused rollout)."""
        owner = getattr(self.cost_function, "__self__", None)
        from .constraint_net import ConstraintNet
        return owner if isinstance(owner, ConstraintNet) and getattr(self.cost_function, "__name__", "") == "cost_function" else None

    def reset(self):
        obs = self.venv.reset()
        self.previous_obs = obs
        return obs

    def step_async(self, actions):
        self.actions = _as_device_f32(actions, self.venv.device) if hasattr(self.venv, "device") else actions
        self.venv.step_async(actions)

    def step_wait(self):
        obs, rews, news, infos = self.venv.step_wait()
        cn = self.constraint_net()
        if cn is not None:
            cost = cn.cost_function_device(self.previous_obs, self.actions)
        else:  # arbitrary Python callable: numpy in / numpy out, as in the reference
            po = self.previous_obs.cpu().numpy() if torch.is_tensor(self.previous_obs) else self.previous_obs
            ac = self.actions.cpu().numpy() if torch.is_tensor(self.actions) else self.actions
            cost = torch.as_tensor(np.asarray(self.cost_function(po.copy(), ac.copy()), dtype=np.float32), device=obs.device)
        infos.batch["cost"] = cost
        self.previous_obs = obs.clone()
        return obs, rews, news, infos


class RunningMeanStd:
    """mean / var / count triple living in HBM (float64).  ``.mean`` / ``.var`` / ``.count`` read back numpy values like
    the reference's attributes (ref: running_mean_std.py:6-18)."""

    def __init__(self, epsilon=1e-4, shape=(), device="cuda"):
        self.shape = tuple(shape)
        n = int(np.prod(shape)) if shape else 1
        self.d_mean = torch.zeros(n, dtype=torch.float64, device=device)
        self.d_var = torch.ones(n, dtype=torch.float64, device=device)
        self.d_count = torch.full((1,), float(epsilon), dtype=torch.float64, device=device)

    @property
    def mean(self):
        return self.d_mean.cpu().numpy().reshape(self.shape).copy() if self.shape else float(self.d_mean.item())

    @property
    def var(self):
        return self.d_var.cpu().numpy().reshape(self.shape).copy() if self.shape else float(self.d_var.item())

    @property
    def count(self):
        return float(self.d_count.item())

    def assign(self, mean, var, count):
        self.d_mean.copy_(torch.as_tensor(np.asarray(mean, np.float64).reshape(-1)))
        self.d_var.copy_(torch.as_tensor(np.asarray(var, np.float64).reshape(-1)))
        self.d_count.fill_(float(count))

    def clone(self):
        c = RunningMeanStd(shape=self.shape, device=self.d_mean.device)
        c.d_mean.copy_(self.d_mean); c.d_var.copy_(self.d_var); c.d_count.copy_(self.d_count)
        return c

    def __deepcopy__(self, memo):
        return self.clone()


class _ScalarRms(RunningMeanStd):
    """ret_rms / cost_rms packed as [mean, var, count] in one device array (the layout icrl_norm_t wants)."""

    def __init__(self, epsilon=1e-4, device="cuda"):
        self.shape = ()
        self.d_stats = torch.tensor([0.0, 1.0, float(epsilon)], dtype=torch.float64, device=device)

    mean = property(lambda self: float(self.d_stats[0].item()))
    var = property(lambda self: float(self.d_stats[1].item()))
    count = property(lambda self: float(self.d_stats[2].item()))

    def assign(self, mean, var, count):
        self.d_stats.copy_(torch.tensor([float(mean), float(var), float(count)], dtype=torch.float64))

    def clone(self):
        c = _ScalarRms(device=self.d_stats.device)
        c.d_stats.copy_(self.d_stats)
        return c


class VecNormalize(VecEnvWrapper):
    """ref: vec_normalize.py:9-181 (obs + reward).  See VecNormalizeWithCost for the cost channel."""

    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0, gamma=0.99,
                 epsilon=1e-8):
        super().__init__(venv)
        dev = self.device = getattr(venv.unwrapped, "device", torch.device("cuda"))
        o = self.observation_space.shape[0]
        self.obs_rms = RunningMeanStd(shape=(o,), device=dev)
        self.ret_rms = _ScalarRms(device=dev)
        self.cost_rms = _ScalarRms(device=dev)
        self.clip_obs, self.clip_reward, self.clip_cost = clip_obs, clip_reward, 10.0
        self.ret = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)
        self.cost_ret = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)
        self.gamma, self.cost_gamma, self.epsilon = gamma, 0.99, epsilon
        self.training, self.norm_obs, self.norm_reward, self.norm_cost = training, norm_obs, norm_reward, False
        self.old_obs = self.old_reward = self.old_cost = None
        self._obs_out = torch.zeros(self.num_envs, o, dtype=torch.float64, device=dev)
        self._rew_out = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)
        self._cost_out = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)

    def struct(self):
        return NormT(int(self.training), int(self.norm_obs), int(self.norm_reward), int(self.norm_cost),
                     float(self.clip_obs), float(self.clip_reward), float(self.clip_cost), float(self.gamma),
                     float(self.cost_gamma), float(self.epsilon), p(self.obs_rms.d_mean), p(self.obs_rms.d_var),
                     p(self.obs_rms.d_count), p(self.ret_rms.d_stats), p(self.cost_rms.d_stats), p(self.ret), p(self.cost_ret))

    def _norm_call(self, obs, rews, cost, news):
        nm = self.struct()
        _lib.check(_lib.lib().icrl_vecnorm_step(_lib.byref(nm), p(obs), p(rews), p(cost), p(news), self.num_envs,
                                                obs.shape[1], p(self._obs_out), p(self._rew_out),
                                                p(self._cost_out) if cost is not None else None, _lib.current_stream()),
                   "icrl_vecnorm_step")

    def step_wait(self):
        obs, rews, news, infos = self.venv.step_wait()
        self.old_obs, self.old_reward = obs, rews
        cost = infos.batch.get(getattr(self, "cost_str", "cost")) if isinstance(infos, BatchedInfos) else None
        if cost is not None:
            self.old_cost = cost
        self._norm_call(obs.contiguous(), rews.contiguous(), None if cost is None else cost.contiguous(), news.contiguous())
        if cost is not None:
            infos.batch[getattr(self, "cost_str", "cost")] = self._cost_out.clone()
        return self._obs_out.clone(), self._rew_out.clone(), news, infos

    def normalize_obs(self, obs):
        """ref: vec_normalize.py:107-114 (no statistics update)."""
        if not self.norm_obs:
            return obs
        o = torch.as_tensor(obs, dtype=torch.float64, device=self.device)
        return torch.clamp((o - self.obs_rms.d_mean) / torch.sqrt(self.obs_rms.d_var + self.epsilon), -self.clip_obs, self.clip_obs)

    def get_original_obs(self):
        return self.old_obs.clone()

    def get_original_reward(self):
        return self.old_reward.clone()

    def reset(self):
        obs = self.venv.reset()
        self.old_obs = obs
        nm = self.struct()
        _lib.check(_lib.lib().icrl_vecnorm_reset(_lib.byref(nm), p(obs.contiguous()), self.num_envs, obs.shape[1],
                                                 p(self._obs_out), _lib.current_stream()), "icrl_vecnorm_reset")
        return self._obs_out.clone()

    # -- persistence: statistics only, like the reference's pickled wrapper minus venv / ret (vec_normalize.py:42-53,159-181)
    def _state(self):
        return dict(obs_rms=(self.obs_rms.mean, self.obs_rms.var, self.obs_rms.count),
                    ret_rms=(self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count),
                    cost_rms=(self.cost_rms.mean, self.cost_rms.var, self.cost_rms.count),
                    **{k: getattr(self, k) for k in ("clip_obs", "clip_reward", "clip_cost", "gamma", "cost_gamma", "epsilon",
                                                      "training", "norm_obs", "norm_reward", "norm_cost")})

    def save(self, save_path):
        with open(save_path, "wb") as f:
            pickle.dump(self._state(), f)

    @classmethod
    def load(cls, load_path, venv):
        """reads the statistics file written by save() or by the REFERENCE (a pickled VecNormalize[WithCost] object,
        vec_normalize.py:42-64,159-181: its class instances are read as attribute bags, no stable_baselines3 / gym needed)."""
        from .utils import load_reference_pickle
        st = load_reference_pickle(load_path)
        if not isinstance(st, dict):
            ref = st.__dict__
            rms = lambda r: (np.asarray(r.mean), np.asarray(r.var), float(r.count))
            st = dict(obs_rms=rms(ref["obs_rms"]), ret_rms=rms(ref["ret_rms"]),
                      cost_rms=rms(ref["cost_rms"]) if "cost_rms" in ref else (0.0, 1.0, 1e-4))
            for k in ("clip_obs", "clip_reward", "clip_cost", "gamma", "cost_gamma", "epsilon", "training", "norm_obs", "norm_reward",
                      "norm_cost", "cost_str"):
                if k in ref:
                    st[k] = ref[k]
        obj = cls(venv)
        for k in ("obs_rms", "ret_rms", "cost_rms"):
            getattr(obj, k).assign(*st.pop(k))
        for k, v in st.items():
            setattr(obj, k, v)
        return obj


class VecNormalizeWithCost(VecNormalize):
    """ref: vec_normalize.py:184-278."""

    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, norm_cost=True, cost_info_str="cost",
                 clip_obs=10.0, clip_reward=10.0, clip_cost=10.0, reward_gamma=0.99, cost_gamma=0.99, epsilon=1e-8):
        super().__init__(venv, training, norm_obs, norm_reward, clip_obs, clip_reward, reward_gamma, epsilon)
        self.norm_cost, self.cost_str, self.clip_cost, self.cost_gamma = norm_cost, cost_info_str, clip_cost, cost_gamma

    def get_original_cost(self):
        return self.old_cost.clone()


def sync_envs_normalization(env, eval_env):
    """obs_rms and ret_rms are copied, cost_rms is not (ref: vec_env/__init__.py:50-65)."""
    env_tmp, eval_tmp = env, eval_env
    while isinstance(env_tmp, VecEnvWrapper):
        if isinstance(env_tmp, VecNormalize):
            eval_tmp.obs_rms = copy.deepcopy(env_tmp.obs_rms)
            eval_tmp.ret_rms = copy.deepcopy(env_tmp.ret_rms)
        env_tmp = env_tmp.venv
        if isinstance(env_tmp, VecCostWrapper):
            env_tmp = env_tmp.venv
        eval_tmp = eval_tmp.venv if isinstance(eval_tmp, VecEnvWrapper) else eval_tmp
