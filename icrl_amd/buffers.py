"""RolloutBufferWithCost — the 16 [T, N, ...] float32 arrays of the reference, resident in HBM.

ref: stable_baselines3/common/buffers.py:443-627 (RolloutBufferWithCost), :53-65 (swap_and_flatten),
     stable_baselines3/common/type_aliases.py:53-63 (RolloutBufferWithCostSamples).

Storage stays time-major [T, N, ...] for the whole life of the buffer (coalesced along N for the GAE scan); the
reference's env-major flattening done by ``get()`` is reproduced by index arithmetic (flat i -> env = i // T, t = i % T)
both in ``get()`` below and inside the fused PPO kernel, so minibatch contents are identical.
"""
from collections import namedtuple

import numpy as np
import torch

from . import _lib
from .structs import BufferT, p

RolloutBufferWithCostSamples = namedtuple(
    "RolloutBufferWithCostSamples",
    ["orig_observations", "observations", "actions", "old_log_prob", "old_reward_values", "reward_advantages",
     "reward_returns", "old_cost_values", "cost_advantages", "cost_returns"])

_SCALARS = ("dones", "log_probs", "rewards", "reward_returns", "reward_values", "reward_advantages", "costs", "orig_costs",
            "cost_returns", "cost_values", "cost_advantages")
_OBS = ("observations", "new_observations", "orig_observations", "new_orig_observations")


class RolloutBufferWithCost:
    def __init__(self, buffer_size, observation_space, action_space, device="cuda", reward_gamma=0.99,
                 reward_gae_lambda=1, cost_gamma=0.99, cost_gae_lambda=1, n_envs=1):
        from . import spaces
        self.buffer_size, self.n_envs = buffer_size, n_envs
        self.observation_space, self.action_space = observation_space, action_space
        self.obs_shape = tuple(observation_space.shape)
        self.action_dim = 1 if isinstance(action_space, spaces.Discrete) else int(action_space.shape[0])
        self.device = torch.device("cuda" if device in ("cpu", "auto", None) else device)
        self.reward_gamma, self.reward_gae_lambda = reward_gamma, reward_gae_lambda
        self.cost_gamma, self.cost_gae_lambda = cost_gamma, cost_gae_lambda
        T, N, dev = buffer_size, n_envs, self.device
        for k in _OBS:
            setattr(self, k, torch.zeros((T, N) + self.obs_shape, device=dev))
        self.actions = torch.zeros(T, N, self.action_dim, device=dev)
        for k in _SCALARS:
            setattr(self, k, torch.zeros(T, N, device=dev))
        # workspace of the GAE launch (icrl_gae_dual_ws): written by GAE launches only; its last 4 bytes are the launch's status word
        self.gae_ws = torch.zeros(int(_lib.lib().icrl_gae_dual_ws_bytes(T, N)) // 8, dtype=torch.int64, device=dev)
        self.gae_status = self.gae_ws.view(torch.int32)[-1:]
        self.pos, self.full, self.generator_ready = 0, False, False

    def struct(self):
        return BufferT(self.buffer_size, self.n_envs, int(np.prod(self.obs_shape)), self.action_dim,
                       p(self.observations), p(self.new_observations), p(self.orig_observations), p(self.new_orig_observations),
                       p(self.actions), p(self.dones), p(self.log_probs), p(self.rewards), p(self.reward_values), p(self.costs),
                       p(self.orig_costs), p(self.cost_values), p(self.reward_advantages), p(self.reward_returns),
                       p(self.cost_advantages), p(self.cost_returns), p(self.gae_ws), self.gae_ws.numel() * 8)

    def reset(self):
        """ref: buffers.py:468-491 (the reference re-allocates zeros; here the arrays are zeroed in place)."""
        for k in _OBS + _SCALARS + ("actions",):
            getattr(self, k).zero_()
        self.pos, self.full, self.generator_ready = 0, False, False

    def add(self, obs, orig_obs, new_obs, new_orig_obs, action, reward, cost, orig_cost, done, reward_value, cost_value, log_prob):
        """ref: buffers.py:554-592 (generic per-step path; the fused rollout writes the same rows from the kernels)."""
        t, dev = self.pos, self.device
        cv = lambda x: torch.as_tensor(x, device=dev).to(torch.float32)
        self.observations[t] = cv(obs).reshape(self.n_envs, -1)
        self.orig_observations[t] = cv(orig_obs).reshape(self.n_envs, -1)
        self.new_observations[t] = cv(new_obs).reshape(self.n_envs, -1)
        self.new_orig_observations[t] = cv(new_orig_obs).reshape(self.n_envs, -1)
        self.actions[t] = cv(action).reshape(self.n_envs, -1)
        self.dones[t] = cv(done).flatten()
        self.log_probs[t] = cv(log_prob).flatten()
        self.rewards[t] = cv(reward).flatten()
        self.reward_values[t] = cv(reward_value).flatten()
        self.costs[t] = cv(cost).flatten()
        self.orig_costs[t] = cv(orig_cost).flatten()
        self.cost_values[t] = cv(cost_value).flatten()
        self.pos += 1
        if self.pos == self.buffer_size:
            self.full = True

    def compute_returns_and_advantage(self, reward_last_value, cost_last_value, dones):
        """ref: buffers.py:543-552 -> icrl_gae_dual (one launch, float64 scan, float32 I/O)."""
        dev = self.device
        lvr = torch.as_tensor(reward_last_value, device=dev).float().flatten().contiguous()
        lvc = torch.as_tensor(cost_last_value, device=dev).float().flatten().contiguous()
        ld = torch.as_tensor(dones, device=dev).to(torch.uint8).flatten().contiguous()
        _lib.check(_lib.lib().icrl_gae_dual_ws(
            p(self.rewards), p(self.costs), p(self.reward_values), p(self.cost_values), p(self.dones), p(lvr), p(lvc), p(ld),
            p(self.reward_advantages), p(self.cost_advantages), p(self.reward_returns), p(self.cost_returns),
            self.buffer_size, self.n_envs, float(self.reward_gamma), float(self.reward_gae_lambda), float(self.cost_gamma),
            float(self.cost_gae_lambda), int(getattr(self, "gae_shape", 0)), p(self.gae_ws), self.gae_ws.numel() * 8,
            _lib.current_stream()), "icrl_gae_dual")

    def check_gae_status(self):
        """raise if a GAE launch reported an expired wait (a workgroup's map never arrived): advantages / returns are then invalid."""
        if int(self.gae_status.item()) != 0:
            self.gae_status.zero_()
            raise RuntimeError("icrl_gae_dual: a wait for another workgroup's affine map expired; advantages and returns are invalid")

    # ---- reference-compatible sampling ---------------------------------------------------------------------------------
    def env_major(self, name):
        """[T,N,...] -> [N*T,...] with flat index env*T + t (ref: buffers.py:53-65)."""
        x = getattr(self, name)
        x = x if x.dim() == 3 else x.unsqueeze(-1)
        return x.transpose(0, 1).reshape(self.buffer_size * self.n_envs, -1)

    def get(self, batch_size=None):
        """ref: buffers.py:594-612 — one np.random.permutation per call, consecutive slices of batch_size."""
        assert self.full, ""
        n = self.buffer_size * self.n_envs
        indices = np.random.permutation(n)
        if batch_size is None:
            batch_size = n
        flat = {k: self.env_major(k) for k in ("orig_observations", "observations", "actions", "log_probs", "reward_values",
                                                "reward_advantages", "reward_returns", "cost_values", "cost_advantages",
                                                "cost_returns")}
        start = 0
        while start < n:
            b = torch.as_tensor(indices[start:start + batch_size], device=self.device)
            yield RolloutBufferWithCostSamples(
                flat["orig_observations"][b], flat["observations"][b], flat["actions"][b], flat["log_probs"][b].flatten(),
                flat["reward_values"][b].flatten(), flat["reward_advantages"][b].flatten(), flat["reward_returns"][b].flatten(),
                flat["cost_values"][b].flatten(), flat["cost_advantages"][b].flatten(), flat["cost_returns"][b].flatten())
            start += batch_size
