"""ActorTwoCriticsPolicy with its parameters in ONE flat float32 HBM buffer.

ref: stable_baselines3/common/policies.py:598-779 (ActorTwoCriticsPolicy), common/torch_layers.py:129-254 (MlpExtractor),
     common/distributions.py:114-192,249-298 (DiagGaussian / Categorical), icrl/utils.py:636-655 (get_net_arch).

Scope: the two-critics MLP policy — an optional shared tanh trunk, then three tanh MLPs (pi / vf / cvf) and a DiagGaussian (Box) or
Categorical (Discrete) head.  Three storage kinds, by architecture:
  * "fast": no trunk, two hidden layers per branch, every width <= 64 (every BASELINE config) — the persistent kernels; narrower
    layers are stored zero-padded to 64;
  * "wide": the same shape with a layer of 65..256 units — generic-shape path, stored zero-padded to a common multiple of 64;
  * "arch": anything else MlpExtractor builds (`-sl` trunk of up to 4 layers, branches of 0..4 layers, widths 1..256) — generic-shape
    path, natural unpadded layout, described to the library by an `arch` array (include/icrl_hip.h: icrl_policy_t).
Parameter initialisation runs on the host with torch-CPU exactly like the reference (same construction / orthogonal-init order, so
the same seed gives the same weights); after that the flat buffer lives on the device and only kernels touch it.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, spaces
from .structs import PolicyT, p

BRANCHES = ("policy_net", "value_net", "cost_value_net")
HW = 64        # width the fast kernels are built for (HD in csrc/ppo_common.h, MAX_H in csrc/common.h)
HW_MAX = 256   # widest layer of the generic-shape path (csrc/generic.hip: GEN_MAX_H)
DEPTH_MAX = 4  # layers of the shared trunk / of one branch on the generic-shape path (csrc/generic.hip: GEN_MAX_DEPTH)


def state_dict_names(discrete, n_hidden=2, n_shared=0):
    """parameter names in the reference's state_dict order; n_hidden: one depth for the three branches or a (pi, vf, cvf) triple."""
    depths = (n_hidden,) * 3 if isinstance(n_hidden, int) else tuple(n_hidden)
    names = [] if discrete else ["log_std"]
    for k in range(n_shared):
        names += [f"mlp_extractor.shared_net.{2 * k}.weight", f"mlp_extractor.shared_net.{2 * k}.bias"]
    for b, d in zip(BRANCHES, depths):
        for k in range(d):
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
        # net_arch = [shared widths..., dict(pi=[...], vf=[...], cvf=[...])] (ref: torch_layers.py:129-254, icrl/utils.py:636-655); a list
        # without the dict is a trunk the three heads read directly
        n_sh = next((i for i, x in enumerate(net_arch) if isinstance(x, dict)), len(net_arch))
        arch = net_arch[n_sh] if n_sh < len(net_arch) else {}
        self.shared = tuple(int(w) for w in net_arch[:n_sh])
        self.layers = {b: tuple(int(w) for w in (arch.get(k) or ())) for b, k in zip(BRANCHES, ("pi", "vf", "cvf"))}
        every = list(self.shared) + [w for b in BRANCHES for w in self.layers[b]]
        if len(self.shared) > DEPTH_MAX or any(len(v) > DEPTH_MAX for v in self.layers.values()) or any(w < 1 or w > HW_MAX for w in every):
            raise NotImplementedError(f"icrl_amd serves net_arch = [shared..., dict(pi=[...], vf=[...], cvf=[...])] with up to {DEPTH_MAX} shared layers, up to "
                                      f"{DEPTH_MAX} layers per branch and 1..{HW_MAX} units per layer; got {net_arch}")
        two = not self.shared and all(len(v) == 2 for v in self.layers.values())
        # logical widths per branch (-pl / -rvl / -cvl).  "fast" / "wide": the device buffers are `hw` wide — 64 when every layer fits the
        # fast kernels, else the widest layer rounded up to a multiple of 64 (the generic-shape path, csrc/generic.hip) — and a narrower
        # layer is stored zero-padded.  Padding units have zero weights in and out and a zero bias, so they output tanh(0) = 0 exactly,
        # receive a zero gradient and keep zero Adam moments: every real parameter sees exactly the arithmetic of the unpadded network
        # (only zeros are added to its dot products).  "arch": natural widths, no padding.
        widest = max(every) if every else 0
        self.kind = ("fast" if widest <= HW else "wide") if two else "arch"
        self.widths = {b: self.layers[b] for b in BRANCHES} if two else None
        self.hw = (HW if widest <= HW else -(-widest // 64) * 64) if two else 0
        self.wide = self.kind != "fast"   # the persistent rollout / sampling / update kernels do not serve this policy
        self.h1 = self.h2 = self.hw
        self.arch = None
        if self.kind == "arch":           # the descriptor icrl_policy_t.arch points at (host memory, kept alive here)
            desc = [len(self.shared), *self.shared]
            for b in BRANCHES:
                desc += [len(self.layers[b]), *self.layers[b]]
            self.arch = np.ascontiguousarray(desc, dtype=np.int32)
        self.optimizer_kwargs = dict(eps=1e-5) if optimizer_kwargs is None else dict(optimizer_kwargs)  # ref: policies.py:357-361
        self.lr_schedule = lr_schedule
        sd = self._init_host(log_std_init, ortho_init)
        self.logical_shapes = OrderedDict((k, tuple(v.shape)) for k, v in sd.items())
        self.shapes = OrderedDict((k, self._physical_shape(k, shp)) for k, shp in self.logical_shapes.items())
        flat = torch.cat([self._pad(k, v).reshape(-1) for k, v in sd.items()]).float()
        self.n_params = flat.numel()
        self.params = flat.to(self.device).contiguous()
        self.params_t = torch.empty_like(self.params)
        # Adam state of policy.optimizer (exp_avg, exp_avg_sq, step) — flat, same layout
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.adam_step = 0
        self.prepare()

    def _physical_shape(self, name, shp):
        if name == "log_std" or self.kind == "arch":
            return shp
        if name.startswith("mlp_extractor."):
            first = name.split(".")[2] == "0"
            return ((self.hw, self.obs_dim if first else self.hw) if name.endswith("weight") else (self.hw,))
        return (shp[0], self.hw) if name.endswith("weight") else shp          # heads: [outputs, last hidden]

    def _pad(self, name, t):
        """logical tensor -> device layout (zeros beyond the logical widths)."""
        t = torch.as_tensor(np.asarray(t), dtype=torch.float32) if not torch.is_tensor(t) else t.detach().float().cpu()
        assert tuple(t.shape) == self.logical_shapes[name], (name, tuple(t.shape), self.logical_shapes[name])
        if self.shapes[name] == self.logical_shapes[name]:
            return t
        out = torch.zeros(self.shapes[name])
        out[tuple(slice(0, n) for n in t.shape)] = t
        return out

    def _split(self, flat, pad=False):
        """flat device-layout vector -> OrderedDict of logical tensors (the padding is dropped)."""
        out, off = OrderedDict(), 0
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            out[k] = flat[off:off + n].reshape(shp)[tuple(slice(0, m) for m in self.logical_shapes[k])].clone()
            off += n
        return out

    def _init_host(self, log_std_init, ortho_init):
        """construction + init order of ActorTwoCriticsPolicy._build (ref: policies.py:648-714, torch_layers.py:208-226)."""
        Lin = torch.nn.Linear
        shared, last_sh = [], self.obs_dim
        for w in self.shared:             # the trunk is built first, then layer k of pi, vf, cvf back to back (zip_longest)
            shared.append(Lin(last_sh, w)); last_sh = w
        lins = {b: [] for b in BRANCHES}
        last = {b: last_sh for b in BRANCHES}
        for k in range(max(len(v) for v in self.layers.values())):
            for b in BRANCHES:
                if k < len(self.layers[b]):
                    lins[b].append(Lin(last[b], self.layers[b][k]))
                    last[b] = self.layers[b][k]
        heads = OrderedDict(action_net=Lin(last["policy_net"], self.act_dim), value_net=Lin(last["value_net"], 1),
                            cost_value_net=Lin(last["cost_value_net"], 1))
        if ortho_init:                    # module.apply order: shared_net, policy_net, value_net, cost_value_net, then the heads
            for lin in shared + [lin for b in BRANCHES for lin in lins[b]]:
                torch.nn.init.orthogonal_(lin.weight, gain=math.sqrt(2)); lin.bias.data.fill_(0.0)
            for lin, g in zip(heads.values(), (0.01, 1.0, 1.0)):
                torch.nn.init.orthogonal_(lin.weight, gain=g); lin.bias.data.fill_(0.0)
        sd = OrderedDict()
        if not self.discrete:
            sd["log_std"] = torch.ones(self.act_dim) * log_std_init
        for k, lin in enumerate(shared):
            sd[f"mlp_extractor.shared_net.{2 * k}.weight"] = lin.weight.data
            sd[f"mlp_extractor.shared_net.{2 * k}.bias"] = lin.bias.data
        for b in BRANCHES:
            for k, lin in enumerate(lins[b]):
                sd[f"mlp_extractor.{b}.{2 * k}.weight"] = lin.weight.data
                sd[f"mlp_extractor.{b}.{2 * k}.bias"] = lin.bias.data
        for name, lin in heads.items():
            sd[f"{name}.weight"], sd[f"{name}.bias"] = lin.weight.data, lin.bias.data
        return sd

    # ---- C-ABI descriptor -------------------------------------------------------------------------------------
    def struct(self):
        return PolicyT(self.obs_dim, self.act_dim, self.h1, self.h2, int(self.discrete), self.n_params, p(self.params), p(self.params_t),
                       None if self.arch is None else self.arch.ctypes.data)

    @property
    def row_floats(self):
        """outputs of every layer of one row: the `row_floats` argument of ICRL_PPO_GENERIC_BYTES (include/icrl_hip.h)."""
        if self.kind == "arch":
            return sum(self.shared) + sum(sum(v) for v in self.layers.values()) + self.act_dim + 2
        return 6 * self.hw + self.act_dim + 2

    def prepare(self):
        """refresh the transposed weight copy the rollout kernels read (call after params change)."""
        s = self.struct()
        _lib.check(_lib.lib().icrl_policy_prepare(_lib.byref(s), _lib.current_stream()), "icrl_policy_prepare")

    # ---- state dict (reference names; ref: expert_data/*/files/best_model.zip:policy.pth) --------------------------
    def state_dict(self):
        return self._split(self.params.detach().cpu())

    def load_state_dict(self, sd):
        flat = torch.cat([self._pad(k, sd[k]).reshape(-1) for k in self.shapes])
        assert flat.numel() == self.n_params
        self.params.copy_(flat.to(self.device))
        self.prepare()

    def optimizer_state_dict(self, lr=None, eps=1e-5):
        """policy.optimizer.state_dict() in torch.optim.Adam's own layout (what SB3 stores as policy.optimizer.pth):
        parameters numbered in state_dict order, per-parameter step / exp_avg / exp_avg_sq."""
        m, v = self._split(self.exp_avg.detach().cpu()), self._split(self.exp_avg_sq.detach().cpu())
        state = {i: dict(step=torch.tensor(float(self.adam_step)), exp_avg=m[k], exp_avg_sq=v[k]) for i, k in enumerate(self.shapes)}
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
        m = torch.cat([self._pad(name, state[k]["exp_avg"]).reshape(-1) for name, k in zip(self.shapes, keys)])
        v = torch.cat([self._pad(name, state[k]["exp_avg_sq"]).reshape(-1) for name, k in zip(self.shapes, keys)])
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
