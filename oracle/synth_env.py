"""Oracle: the synthetic HCWithPos- / AntWall-shaped vectorised environment.  Test infrastructure only.

MuJoCo is unavailable on either box, and BASELINE.json quotes the metric on "synthetic
HCWithPos-shaped obs/action tensors", so both the CPU baseline and the HIP path step this
environment.  It is *our* definition, not the reference's; what it borrows
from the reference is the contract:

  * shapes / episode lengths: HC obs 18, act 6, 1000 steps (ref: custom_envs/envs/half_cheetah.py:138-144,
    custom_envs/__init__.py:43-49); Ant obs 113, act 8, 500 steps (ref: custom_envs/__init__.py:194-216);
  * reward form: |dx|/dt - 0.1*|a|^2 (ref: half_cheetah.py:152-155) or |xy| + 1 - 0.5*|a|^2 (ref: ant.py:62-75);
  * done only at the time limit, auto-reset by the vec-env (ref: subproc_vec_env.py:20-26);
    "Test" variants also terminate with reward 0 when obs[0] <= -3 (ref: half_cheetah.py:196-221, ant.py:95-102);
  * AntWallBroken zeroes action[4:] (ref: ant.py:105-108);
  * env i is seeded ``seed + i`` (ref: icrl/utils.py:256-263).

Dynamics, all float64 with one rounding per operation and NO fused multiply-add, in this order:

    acc_i  = 0.99 * s_i
    acc_i  = acc_i + B[i][j] * a_j          for j = 0 .. act-1
    s'_i   = acc_i + 0.01 * eps_i
    eps_i  = (u24(key, step_count, i) / 2^24 - 0.5) * sqrt(12)        (unit-variance uniform)
    s0_i   = (u24(key, step_count, obs+i) / 2^24 - 0.5) * 0.2         (reset draw)

``u24`` is a counter-based 24-bit hash (three murmur3 finalisers) so the stream of env i does not
depend on how many envs are stepped together or on which device steps them.
"""
import numpy as np

F64 = np.float64
M32 = np.uint64(0xFFFFFFFF)
SQRT12 = 3.4641016151377544

KINDS = {
    # kind: (obs_dim, act_dim, max_episode_steps, reward form)
    "hc": (18, 6, 1000, 0),
    "ant": (113, 8, 500, 1),
}


def _fmix32(x):
    x = x & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M32
    x ^= x >> np.uint64(16)
    return x


def u24(key, ctr, comp):
    """key, ctr, comp: broadcastable uint64 arrays holding 32-bit values -> uint64 in [0, 2^24)."""
    x = _fmix32(key ^ np.uint64(0x9E3779B9))
    x = _fmix32((x + ((ctr & M32) * np.uint64(0x9E3779B1) & M32)) & M32)
    x = _fmix32(x ^ ((comp * np.uint64(0x7FEB352D)) & M32))
    return x >> np.uint64(8)


def unit_uniform(key, ctr, comp):
    return u24(key, ctr, comp).astype(F64) / F64(16777216.0)


def dynamics_matrix(kind):
    obs_dim, act_dim, _, _ = KINDS[kind]
    return (np.random.RandomState(1234).randn(obs_dim, act_dim) * 0.05).astype(F64)


class SynthVecEnv:
    """Vectorised numpy implementation (the CPU port steps this inside a Python loop)."""

    def __init__(self, n_envs, kind="hc", seed=0, env_index_offset=0, wall_terminate=False, broken=False):
        self.kind = kind
        self.obs_dim, self.act_dim, self.max_steps, self.reward_form = KINDS[kind]
        self.n_envs = n_envs
        self.B = dynamics_matrix(kind)
        self.wall_terminate = wall_terminate
        self.broken = broken
        self.action_low = -np.ones(self.act_dim, np.float32)
        self.action_high = np.ones(self.act_dim, np.float32)
        self.seed(seed, env_index_offset)

    def seed(self, seed, env_index_offset=0):
        self.key = (np.arange(self.n_envs, dtype=np.uint64) + np.uint64(seed + env_index_offset)) & M32
        self.step_count = np.zeros(self.n_envs, np.uint64)
        self.t_ep = np.zeros(self.n_envs, np.int64)
        self.s = np.zeros((self.n_envs, self.obs_dim), F64)

    def _draw_s0(self, idx):
        comp = np.arange(self.obs_dim, dtype=np.uint64)[None, :] + np.uint64(self.obs_dim)
        u = unit_uniform(self.key[idx, None], self.step_count[idx, None], comp)
        return (u - 0.5) * 0.2

    def reset(self):
        idx = np.arange(self.n_envs)
        self.s[idx] = self._draw_s0(idx)
        self.t_ep[:] = 0
        return self.s.copy()

    def step(self, actions):
        a = np.asarray(actions).astype(F64).reshape(self.n_envs, self.act_dim).copy()
        if self.broken:
            a[:, 4:] = 0.0
        comp = np.arange(self.obs_dim, dtype=np.uint64)[None, :]
        eps = (unit_uniform(self.key[:, None], self.step_count[:, None], comp) - 0.5) * SQRT12
        old0 = self.s[:, 0].copy()
        acc = 0.99 * self.s
        sq = np.zeros(self.n_envs, F64)
        for j in range(self.act_dim):
            acc = acc + self.B[None, :, j] * a[:, j, None]
            sq = sq + a[:, j] * a[:, j]
        new_s = acc + 0.01 * eps
        if self.reward_form == 0:
            rew = np.abs(new_s[:, 0] - old0) / 0.05 - 0.1 * sq
        else:
            rew = (np.sqrt(new_s[:, 0] * new_s[:, 0] + new_s[:, 1] * new_s[:, 1]) + 1.0) - 0.5 * sq
        done = np.zeros(self.n_envs, bool)
        if self.wall_terminate:
            hit = new_s[:, 0] <= -3.0
            rew = np.where(hit, 0.0, rew)
            done |= hit
        self.t_ep += 1
        self.step_count += np.uint64(1)
        done |= self.t_ep >= self.max_steps
        self.s = new_s
        if done.any():
            idx = np.nonzero(done)[0]
            self.s[idx] = self._draw_s0(idx)
            self.t_ep[idx] = 0
        return self.s.copy(), rew, done
