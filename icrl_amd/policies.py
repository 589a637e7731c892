"""ActorTwoCriticsPolicy with its parameters in ONE flat float32 HBM buffer.

ref: stable_baselines3/common/policies.py:598-779 (ActorTwoCriticsPolicy), common/torch_layers.py:129-254 (MlpExtractor),
     common/distributions.py:114-192,249-298 (DiagGaussian / Categorical), icrl/utils.py:636-655 (get_net_arch).

Scope: the two-critics MLP policy every BASELINE config uses — three separate tanh MLPs (pi / vf / cvf) with two hidden
layers each (<= 64 wide), no shared trunk, DiagGaussian (Box) or Categorical (Discrete) head.  Parameter initialisation
runs on the host with torch-CPU exactly like the reference (same construction / orthogonal-init order, so the same seed
gives the same weights); after that the flat buffer lives on the device and only kernels touch it.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, spaces
from .structs import PolicyT, p

BRANCHES = ("policy_net", "value_net", "cost_value_net")


def state_dict_names(discrete, n_hidden=2):
    names = [] if discrete else ["log_std"]
    for b in BRANCHES:
        for k in range(n_hidden):
            names += [f"mlp_extractor.{b}.{2 * k}.weight", f"mlp_extractor.{b}.{2 * k}.bias"]
    for h in ("action_net", "value_net", "cost_value_net"):
        names += [f"{h}.weight", f"{h}.bias"]
    return names


class ActorTwoCriticsPolicy:
    def __init__(self, observation_space, action_space, lr_schedule=None, net_arch=None, log_std_init=0.0,
                 ortho_init=True, optimizer_kwargs=None, device="cuda"):
        self.observation_space, self.action_space = observation_space, action_space
        self.device = torch.device(device)
        self.obs_dim = int(observation_space.shape[0])
        self.discrete = isinstance(action_space, spaces.Discrete)
        self.act_dim = int(action_space.n) if self.discrete else int(action_space.shape[0])
        if net_arch is None:
            net_arch = [dict(pi=[64, 64], vf=[64, 64], cvf=[64, 64])]
        arch = net_arch[-1] if isinstance(net_arch[-1], dict) else None
        if arch is None or len(net_arch) != 1 or not (arch.get("pi") == arch.get("vf") == arch.get("cvf")) \
                or len(arch["pi"]) != 2 or max(arch["pi"]) > 64:
            raise NotImplementedError("icrl_amd supports net_arch=[dict(pi=[h1,h2], vf=[h1,h2], cvf=[h1,h2])] with h<=64 "
                                      f"(every BASELINE config); got {net_arch}")
        self.h1, self.h2 = int(arch["pi"][0]), int(arch["pi"][1])
        self.optimizer_kwargs = dict(eps=1e-5) if optimizer_kwargs is None else dict(optimizer_kwargs)  # ref: policies.py:357-361
        self.lr_schedule = lr_schedule
        sd = self._init_host(log_std_init, ortho_init)
        self.shapes = OrderedDict((k, tuple(v.shape)) for k, v in sd.items())
        flat = torch.cat([v.reshape(-1) for v in sd.values()]).float()
        self.n_params = flat.numel()
        self.params = flat.to(self.device).contiguous()
        self.params_t = torch.empty_like(self.params)
        # Adam state of policy.optimizer (exp_avg, exp_avg_sq, step) — flat, same layout
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.adam_step = 0
        self.prepare()

    def _init_host(self, log_std_init, ortho_init):
        """construction + init order of ActorTwoCriticsPolicy._build (ref: policies.py:648-714, torch_layers.py:208-226)."""
        Lin = torch.nn.Linear
        lins = {b: [] for b in BRANCHES}
        last = self.obs_dim
        for h in (self.h1, self.h2):
            for b in BRANCHES:
                lins[b].append(Lin(last, h))
            last = h
        heads = OrderedDict(action_net=Lin(last, self.act_dim), value_net=Lin(last, 1), cost_value_net=Lin(last, 1))
        if ortho_init:
            for b in BRANCHES:
                for lin in lins[b]:
                    torch.nn.init.orthogonal_(lin.weight, gain=math.sqrt(2)); lin.bias.data.fill_(0.0)
            for lin, g in zip(heads.values(), (0.01, 1.0, 1.0)):
                torch.nn.init.orthogonal_(lin.weight, gain=g); lin.bias.data.fill_(0.0)
        sd = OrderedDict()
        if not self.discrete:
            sd["log_std"] = torch.ones(self.act_dim) * log_std_init
        for b in BRANCHES:
            for k, lin in enumerate(lins[b]):
                sd[f"mlp_extractor.{b}.{2 * k}.weight"] = lin.weight.data
                sd[f"mlp_extractor.{b}.{2 * k}.bias"] = lin.bias.data
        for name, lin in heads.items():
            sd[f"{name}.weight"], sd[f"{name}.bias"] = lin.weight.data, lin.bias.data
        return sd

    # ---- C-ABI descriptor -------------------------------------------------------------------------------------
    def struct(self):
        return PolicyT(self.obs_dim, self.act_dim, self.h1, self.h2, int(self.discrete), self.n_params, p(self.params), p(self.params_t))

    def prepare(self):
        """refresh the transposed weight copy the rollout kernels read (call after params change)."""
        s = self.struct()
        _lib.check(_lib.lib().icrl_policy_prepare(_lib.byref(s), _lib.current_stream()), "icrl_policy_prepare")

    # ---- state dict (reference names; ref: expert_data/*/files/best_model.zip:policy.pth) --------------------------
    def state_dict(self):
        out, off = OrderedDict(), 0
        flat = self.params.detach().cpu()
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            out[k] = flat[off:off + n].reshape(shp).clone()
            off += n
        return out

    def load_state_dict(self, sd):
        flat = torch.cat([torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32).reshape(-1) if not torch.is_tensor(sd[k])
                          else sd[k].detach().float().reshape(-1).cpu() for k in self.shapes])
        assert flat.numel() == self.n_params
        self.params.copy_(flat.to(self.device))
        self.prepare()

    def optimizer_state_dict(self, lr=None, eps=1e-5):
        """policy.optimizer.state_dict() in torch.optim.Adam's own layout (what SB3 stores as policy.optimizer.pth):
        parameters numbered in state_dict order, per-parameter step / exp_avg / exp_avg_sq."""
        state, off = {}, 0
        m, v = self.exp_avg.detach().cpu(), self.exp_avg_sq.detach().cpu()
        for i, (k, shp) in enumerate(self.shapes.items()):
            n = int(np.prod(shp))
            state[i] = dict(step=torch.tensor(float(self.adam_step)), exp_avg=m[off:off + n].reshape(shp).clone(),
                            exp_avg_sq=v[off:off + n].reshape(shp).clone())
            off += n
        group = dict(lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=0, amsgrad=False, params=list(range(len(self.shapes))))
        return dict(state=state if self.adam_step > 0 else {}, param_groups=[group])

    def load_optimizer_state_dict(self, osd):
        state = osd.get("state", {})
        if not state:
            self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.adam_step = 0
            return
        # parameter order = param_groups[0]["params"]: 0..n-1 in current torch, the parameters' id()s in the torch 1.5 files the
        # reference wrote (expert_data/*/files/best_model.zip); both follow policy.parameters() = state_dict order
        keys = [k for g in osd.get("param_groups", []) for k in g["params"]] or sorted(state)
        assert len(keys) == len(self.shapes), "optimizer state does not match the policy's parameter list"
        m = torch.cat([state[k]["exp_avg"].detach().float().reshape(-1) for k in keys])
        v = torch.cat([state[k]["exp_avg_sq"].detach().float().reshape(-1) for k in keys])
        assert m.numel() == self.exp_avg.numel()
        self.exp_avg.copy_(m.to(self.device)); self.exp_avg_sq.copy_(v.to(self.device))
        self.adam_step = int(float(state[keys[0]]["step"]))

    @property
    def log_std(self):
        return None if self.discrete else self.params[:self.act_dim]

    # ---- inference ------------------------------------------------------------------------------------------------
    def forward(self, obs, deterministic=False, noise=None, clip=True):
        """ref: policies.py:716-731.  obs: [N,obs] (device float64 or anything convertible).  noise: [N,act] standard normals
        ([N] uniforms when discrete); None draws them on the device.  Returns (actions, reward_values, cost_values, log_prob)
        plus ``self.last_clipped`` = actions clipped to the action box."""
        dev = self.device
        obs = torch.as_tensor(obs, device=dev).to(torch.float64).reshape(-1, self.obs_dim).contiguous()
        n = obs.shape[0]
        if noise is None and not deterministic:
            noise = torch.rand(n, device=dev) if self.discrete else torch.randn(n, self.act_dim, device=dev)
        if noise is not None:
            noise = torch.as_tensor(noise, device=dev).float().contiguous()
        a_store = 1 if self.discrete else self.act_dim
        actions = torch.empty(n, a_store, device=dev); clipped = torch.empty(n, a_store, device=dev)
        v_r, v_c, lp = (torch.empty(n, device=dev) for _ in range(3))
        lo = hi = None
        if clip and not self.discrete:
            lo = torch.as_tensor(self.action_space.low, device=dev).float().contiguous()
            hi = torch.as_tensor(self.action_space.high, device=dev).float().contiguous()
        s = self.struct()
        _lib.check(_lib.lib().icrl_policy_forward(_lib.byref(s), p(obs), p(noise), n, int(deterministic), p(lo), p(hi),
                                                  p(actions), p(clipped), p(v_r), p(v_c), p(lp), _lib.current_stream()),
                   "icrl_policy_forward")
        self.last_clipped = clipped
        return actions, v_r.reshape(-1, 1), v_c.reshape(-1, 1), lp

    def predict(self, observation, state=None, mask=None, deterministic=False, noise=None):
        """ref: policies.py:215-280 — sampled (or mode) action clipped to the action box; returns (actions, state)."""
        self.forward(observation, deterministic, noise)
        return self.last_clipped, state

    def evaluate_actions(self, obs, actions):
        """ref: policies.py:752-767 -> (reward_values, cost_values, log_prob, entropy); device tensors."""
        dev = self.device
        obs = torch.as_tensor(obs, device=dev).to(torch.float64).reshape(-1, self.obs_dim).contiguous()
        a_store = 1 if self.discrete else self.act_dim
        actions = torch.as_tensor(actions, device=dev).to(torch.float32).reshape(-1, a_store).contiguous()
        n = obs.shape[0]
        v_r, v_c, lp, ent = (torch.empty(n, device=dev) for _ in range(4))
        s = self.struct()
        _lib.check(_lib.lib().icrl_policy_evaluate(_lib.byref(s), p(obs), p(actions), n, p(v_r), p(v_c), p(lp), p(ent),
                                                   _lib.current_stream()), "icrl_policy_evaluate")
        return v_r.reshape(-1, 1), v_c.reshape(-1, 1), lp, ent
