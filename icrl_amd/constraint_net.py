"""ConstraintNet (zeta_theta) — same constructor / methods as the reference, arithmetic in libicrl_hip.so.

ref: icrl/constraint_net.py:14-402.  Parameters live in one flat float32 HBM buffer in the reference's state_dict order
(network.0.weight, network.0.bias, ...).  Kept quirks: select_dim appends range(acs_dim) for the
action part (so it re-selects leading *observation* columns); ConstraintNet.load passes constructor arguments one slot off.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .structs import CostNetT, p


class ConstraintNet:
    def __init__(self, obs_dim, acs_dim, hidden_sizes, batch_size, lr_schedule, expert_obs, expert_acs, is_discrete,
                 regularizer_coeff=0., obs_select_dim=None, acs_select_dim=None, optimizer_class=torch.optim.Adam,
                 optimizer_kwargs=None, no_importance_sampling=False, per_step_importance_sampling=False, clip_obs=10.,
                 initial_obs_mean=None, initial_obs_var=None, action_low=None, action_high=None, target_kl_old_new=-1,
                 target_kl_new_old=-1, train_gail_lambda=False, eps=1e-5, device="cuda"):
        self.obs_dim, self.acs_dim = obs_dim, acs_dim
        self.obs_select_dim, self.acs_select_dim = obs_select_dim, acs_select_dim
        self._define_input_dims()
        self.expert_obs, self.expert_acs = expert_obs, expert_acs
        self.hidden_sizes, self.batch_size, self.is_discrete = hidden_sizes, batch_size, is_discrete
        self.regularizer_coeff = regularizer_coeff
        self.importance_sampling = not no_importance_sampling
        self.per_step_importance_sampling = per_step_importance_sampling
        self.clip_obs, self.eps = clip_obs, eps
        self.device = torch.device("cuda" if device in (None, "cpu", "auto") else device)  # HBM-resident regardless
        self.train_gail_lambda = train_gail_lambda
        if optimizer_kwargs is None:
            optimizer_kwargs = {}
            if optimizer_class == torch.optim.Adam:
                optimizer_kwargs["eps"] = 1e-5                      # ref: constraint_net.py:65-69
        self.optimizer_kwargs, self.optimizer_class, self.lr_schedule = optimizer_kwargs, optimizer_class, lr_schedule
        self.current_obs_mean, self.current_obs_var = initial_obs_mean, initial_obs_var
        self.action_low, self.action_high = action_low, action_high
        self.target_kl_old_new, self.target_kl_new_old = target_kl_old_new, target_kl_new_old
        self.current_progress_remaining = 1.
        self._build()

    # ref: constraint_net.py:86-99
    def _define_input_dims(self):
        self.select_dim = []
        if self.obs_select_dim is None:
            self.select_dim += [i for i in range(self.obs_dim)]
        elif self.obs_select_dim[0] != -1:
            self.select_dim += list(self.obs_select_dim)
        if self.acs_select_dim is None:
            self.select_dim += [i for i in range(self.acs_dim)]
        elif self.acs_select_dim[0] != -1:
            self.select_dim += list(self.acs_select_dim)
        assert len(self.select_dim) > 0, ""
        self.input_dims = len(self.select_dim)

    def _build(self):
        """ref: constraint_net.py:101-116 — create_mlp(input, 1, hidden) with ReLU + Sigmoid; host init, device storage."""
        hs = list(self.hidden_sizes)
        if not (1 <= len(hs) <= 2) or max(hs) > 64:
            raise NotImplementedError(f"icrl_amd ConstraintNet supports 1-2 hidden layers of <= 64 units, got {hs}")
        sd, last, k = OrderedDict(), self.input_dims, 0
        for h in hs + [1]:
            lin = torch.nn.Linear(last, h)
            sd[f"{k}.weight"], sd[f"{k}.bias"] = lin.weight.data, lin.bias.data
            last, k = h, k + 2
        self.shapes = OrderedDict((n, tuple(v.shape)) for n, v in sd.items())
        flat = torch.cat([v.reshape(-1) for v in sd.values()]).float()
        self.n_params = flat.numel()
        dev = self.device
        self.params = flat.to(dev).contiguous()
        self.params_t = torch.empty_like(self.params)
        self.exp_avg, self.exp_avg_sq, self.adam_step = torch.zeros_like(self.params), torch.zeros_like(self.params), 0
        self.optimizer = "adam" if self.optimizer_class is not None else None
        self.d_select = torch.as_tensor(np.asarray(self.select_dim, np.int32), device=dev)
        self._refresh_consts()
        self.prepare()

    def _refresh_consts(self):
        dev = self.device
        f32 = lambda x: None if x is None else torch.as_tensor(np.asarray(x, np.float32), device=dev).contiguous()
        f64 = lambda x: None if x is None else torch.as_tensor(np.asarray(x, np.float64), device=dev).contiguous()
        both = self.action_low is not None and self.action_high is not None      # ref: clip_actions, :293-297
        self.d_low, self.d_high = (f32(self.action_low), f32(self.action_high)) if both else (None, None)
        norm = self.current_obs_mean is not None and self.current_obs_var is not None
        self.d_mean, self.d_var = (f64(self.current_obs_mean), f64(self.current_obs_var)) if norm else (None, None)

    def struct(self):
        hs = list(self.hidden_sizes)
        return CostNetT(self.obs_dim, self.acs_dim, self.input_dims, len(hs), hs[0], hs[1] if len(hs) > 1 else 0,
                        int(bool(self.is_discrete)), self.n_params, -1.0 if self.clip_obs is None else float(self.clip_obs),
                        p(self.d_select), p(self.d_low), p(self.d_high), p(self.d_mean), p(self.d_var), float(self.eps),
                        p(self.params), p(self.params_t))

    def prepare(self):
        s = self.struct()
        _lib.check(_lib.lib().icrl_costnet_prepare(_lib.byref(s), _lib.current_stream()), "icrl_costnet_prepare")

    # ---- inference ------------------------------------------------------------------------------------------------
    def cost_function_device(self, obs, acs):
        """device tensors in, device float32 [N] out (what VecCostWrapper uses)."""
        dev = self.device
        obs = torch.as_tensor(obs, device=dev).to(torch.float64).reshape(-1, self.obs_dim).contiguous()
        a_w = 1 if self.is_discrete else self.acs_dim
        acs = torch.as_tensor(acs, device=dev).to(torch.float32).reshape(-1, a_w).contiguous()
        cost = torch.empty(obs.shape[0], device=dev)
        s = self.struct()
        _lib.check(_lib.lib().icrl_cost_mlp_forward(_lib.byref(s), p(obs), p(acs), obs.shape[0], p(cost), _lib.current_stream()),
                   "icrl_cost_mlp_forward")
        return cost

    def cost_function(self, obs, acs):
        """ref: constraint_net.py:121-130 — numpy in, numpy out."""
        assert obs.shape[-1] == self.obs_dim, ""
        if not self.is_discrete:
            assert acs.shape[-1] == self.acs_dim, ""
        lead = obs.shape[:-1]
        return self.cost_function_device(obs, acs).cpu().numpy().reshape(lead)

    # ---- state dict ------------------------------------------------------------------------------------------------
    def state_dict(self):
        out, off, flat = OrderedDict(), 0, self.params.detach().cpu()
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            out[k] = flat[off:off + n].reshape(shp).clone()
            off += n
        return out

    def load_state_dict(self, sd):
        flat = torch.cat([(sd[k].detach().float().cpu() if torch.is_tensor(sd[k]) else torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32)).reshape(-1)
                          for k in self.shapes])
        self.params.copy_(flat.to(self.device))
        self.prepare()
