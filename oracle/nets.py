"""Oracle: actor + two critics policy and the constraint (cost) network, torch-CPU fp32.
Test infrastructure only.

ref: stable_baselines3/common/policies.py:598-779      (ActorTwoCriticsPolicy)
     stable_baselines3/common/torch_layers.py:93-254    (create_mlp, MlpExtractor)
     stable_baselines3/common/distributions.py:114-192  (DiagGaussianDistribution)
     stable_baselines3/common/distributions.py:249-298  (CategoricalDistribution)
     icrl/constraint_net.py:101-130,258-299             (ConstraintNet._build / cost_function / prepare_data)

Third-party arithmetic: torch (here 2.10.0; the reference pins 1.5.0) supplies Linear/tanh/
ReLU/sigmoid, autograd, Adam and clip_grad_norm_.  The oracle calls the same torch ops the
reference calls; what is restated here is the *composition* (which nets, which order, which
formulas for log-prob / entropy / clipping).

Parameters are kept in a dict keyed by the names of the reference's state_dict so that a
``policy.pth`` from the reference loads directly (19 tensors for the two-critics MLP policy).
"""
import math
from collections import OrderedDict

import numpy as np
import torch as th
import torch.nn.functional as F

LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


class TwoCriticPolicy:
    """obs -> (pi | vf | cvf) tanh MLPs -> action head (+ state-independent log_std) and 2 value heads."""

    BRANCHES = ("policy_net", "value_net", "cost_value_net")

    def __init__(self, obs_dim, act_dim, hidden=(64, 64), discrete=False, log_std_init=0.0, ortho_init=True, shared=()):
        # hidden: one width list for the three branches, or a dict branch -> width list of any length (net_arch pi / vf / cvf,
        # icrl/utils.py:636-655); shared: widths of the trunk in front of the branches (-sl)
        widths = {b: tuple(hidden[b]) for b in self.BRANCHES} if isinstance(hidden, dict) else {b: tuple(hidden) for b in self.BRANCHES}
        self.obs_dim, self.act_dim, self.discrete = obs_dim, act_dim, discrete
        self.hidden = widths[self.BRANCHES[0]]
        self.widths, self.shared = widths, tuple(shared)
        # construction order mirrors MlpExtractor (ref: torch_layers.py:183-226): the trunk first, then layer k of pi, vf, cvf back
        # to back (zip_longest), so the torch RNG stream matches.
        trunk, last_sh = [], obs_dim
        for w in self.shared:
            trunk.append(th.nn.Linear(last_sh, w)); last_sh = w
        lins = {b: [] for b in self.BRANCHES}
        last = {b: last_sh for b in self.BRANCHES}
        for k in range(max(len(v) for v in widths.values())):
            for b in self.BRANCHES:
                if k < len(widths[b]):
                    lins[b].append(th.nn.Linear(last[b], widths[b][k]))
                    last[b] = widths[b][k]
        action_net = th.nn.Linear(last["policy_net"], act_dim)
        value_net = th.nn.Linear(last["value_net"], 1)
        cost_value_net = th.nn.Linear(last["cost_value_net"], 1)
        if ortho_init:  # ref: policies.py:697-711 — gains sqrt(2) / 0.01 / 1 / 1, biases zero; module.apply order
            for lin in trunk + [lin for b in self.BRANCHES for lin in lins[b]]:
                th.nn.init.orthogonal_(lin.weight, gain=math.sqrt(2)); lin.bias.data.fill_(0.0)
            for lin, g in ((action_net, 0.01), (value_net, 1.0), (cost_value_net, 1.0)):
                th.nn.init.orthogonal_(lin.weight, gain=g); lin.bias.data.fill_(0.0)
        p = OrderedDict()
        if not discrete:
            p["log_std"] = th.ones(act_dim) * log_std_init
        for k, lin in enumerate(trunk):
            p[f"mlp_extractor.shared_net.{2 * k}.weight"] = lin.weight.data.clone()
            p[f"mlp_extractor.shared_net.{2 * k}.bias"] = lin.bias.data.clone()
        for b in self.BRANCHES:
            for k, lin in enumerate(lins[b]):
                p[f"mlp_extractor.{b}.{2 * k}.weight"] = lin.weight.data.clone()
                p[f"mlp_extractor.{b}.{2 * k}.bias"] = lin.bias.data.clone()
        for name, lin in (("action_net", action_net), ("value_net", value_net), ("cost_value_net", cost_value_net)):
            p[f"{name}.weight"] = lin.weight.data.clone()
            p[f"{name}.bias"] = lin.bias.data.clone()
        self.params = OrderedDict((k, v.requires_grad_(True)) for k, v in p.items())

    # -- state dict ------------------------------------------------------------------------
    def load_state_dict(self, sd):
        for k in self.params:
            self.params[k].data.copy_(th.as_tensor(np.asarray(sd[k])) if not th.is_tensor(sd[k]) else sd[k])

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def parameters(self):
        return list(self.params.values())

    # -- forward ---------------------------------------------------------------------------
    def _branch(self, x, b, depth):
        for k in range(depth):
            x = th.tanh(F.linear(x, self.params[f"mlp_extractor.{b}.{2 * k}.weight"],
                                 self.params[f"mlp_extractor.{b}.{2 * k}.bias"]))
        return x

    def latents(self, obs):
        obs = obs.float()                                   # ref: preprocessing.py:61
        shared_latent = self._branch(obs, "shared_net", len(self.shared))      # ref: torch_layers.py:245-254
        return tuple(self._branch(shared_latent, b, len(self.widths[b])) for b in self.BRANCHES)

    def heads(self, obs):
        lp, lv, lc = self.latents(obs)
        mean = F.linear(lp, self.params["action_net.weight"], self.params["action_net.bias"])
        v_r = F.linear(lv, self.params["value_net.weight"], self.params["value_net.bias"])
        v_c = F.linear(lc, self.params["cost_value_net.weight"], self.params["cost_value_net.bias"])
        return mean, v_r, v_c

    def gaussian_log_prob(self, mean, actions):
        """ref: distributions.py:143-161 -> torch.distributions.Normal.log_prob, summed over action dims."""
        std = th.ones_like(mean) * self.params["log_std"].exp()
        var = std ** 2
        lp = -((actions - mean) ** 2) / (2 * var) - std.log() - LOG_SQRT_2PI
        return lp.sum(dim=1)

    def gaussian_entropy(self, mean):
        std = th.ones_like(mean) * self.params["log_std"].exp()
        return (0.5 + 0.5 * math.log(2 * math.pi) + th.log(std)).sum(dim=1)

    def forward(self, obs, noise=None, deterministic=False):
        """Rollout-time forward (ref: policies.py:716-731).  ``noise`` ([N, act] standard normal,
        or [N] uniforms for discrete actions) teacher-forces the sample; None draws from torch's
        global generator exactly as Normal.rsample does."""
        mean, v_r, v_c = self.heads(obs)
        if self.discrete:
            logp_all = mean - mean.logsumexp(dim=-1, keepdim=True)
            if deterministic:
                actions = th.argmax(logp_all.exp(), dim=1)
            elif noise is None:
                actions = th.distributions.Categorical(logits=mean).sample()
            else:
                cdf = th.cumsum(logp_all.exp(), dim=1)
                actions = (noise.reshape(-1, 1) >= cdf).sum(dim=1).clamp(max=self.act_dim - 1)
            log_prob = logp_all.gather(1, actions.reshape(-1, 1)).squeeze(1)
            return actions, v_r, v_c, log_prob
        if deterministic:
            actions = mean
        else:
            std = th.ones_like(mean) * self.params["log_std"].exp()
            eps = th.randn(mean.shape) if noise is None else noise   # Normal.rsample: loc + eps * scale
            actions = mean + eps * std
        return actions, v_r, v_c, self.gaussian_log_prob(mean, actions)

    def evaluate_actions(self, obs, actions):
        """ref: policies.py:752-767 -> (v_r, v_c, log_prob, entropy)."""
        mean, v_r, v_c = self.heads(obs)
        if self.discrete:
            logp_all = mean - mean.logsumexp(dim=-1, keepdim=True)
            a = actions.long().flatten()
            log_prob = logp_all.gather(1, a.reshape(-1, 1)).squeeze(1)
            p = th.softmax(logp_all, dim=-1)        # torch.distributions.Categorical: probs = softmax(normalised logits)
            entropy = -(logp_all * p).sum(-1)
            return v_r, v_c, log_prob, entropy
        return v_r, v_c, self.gaussian_log_prob(mean, actions), self.gaussian_entropy(mean)


class CostNet:
    """zeta_theta: ReLU MLP + sigmoid over [clip(obs), clip(acs)][select_dim].
    ref: constraint_net.py:101-130 (build / cost_function), :258-299 (prepare_data)."""

    def __init__(self, obs_dim, acs_dim, hidden, is_discrete=False, obs_select_dim=None, acs_select_dim=None,
                 clip_obs=10.0, action_low=None, action_high=None, eps=1e-5):
        self.obs_dim, self.acs_dim, self.hidden, self.is_discrete = obs_dim, acs_dim, tuple(hidden), is_discrete
        sel = []
        if obs_select_dim is None:
            sel += list(range(obs_dim))
        elif obs_select_dim[0] != -1:
            sel += list(obs_select_dim)
        if acs_select_dim is None:
            sel += list(range(acs_dim))          # NB: indexes the concatenated vector, as the reference does
        elif acs_select_dim[0] != -1:
            sel += list(acs_select_dim)
        self.select_dim = sel
        self.clip_obs, self.action_low, self.action_high, self.eps = clip_obs, action_low, action_high, eps
        self.obs_mean = self.obs_var = None
        p = OrderedDict()
        last = len(sel)
        k = 0
        for h in self.hidden:
            lin = th.nn.Linear(last, h)
            p[f"{k}.weight"], p[f"{k}.bias"] = lin.weight.data.clone(), lin.bias.data.clone()
            last, k = h, k + 2
        lin = th.nn.Linear(last, 1)
        p[f"{k}.weight"], p[f"{k}.bias"] = lin.weight.data.clone(), lin.bias.data.clone()
        self.n_layers = len(self.hidden) + 1
        self.params = OrderedDict((key, v.requires_grad_(True)) for key, v in p.items())

    def parameters(self):
        return list(self.params.values())

    def load_state_dict(self, sd):
        for k in self.params:
            self.params[k].data.copy_(sd[k] if th.is_tensor(sd[k]) else th.as_tensor(np.asarray(sd[k])))

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def prepare(self, obs, acs):
        obs = np.asarray(obs)
        if self.obs_mean is not None and self.obs_var is not None:
            obs = (obs - self.obs_mean[None]) / np.sqrt(self.obs_var[None] + self.eps)
        if self.clip_obs is not None:
            obs = np.clip(obs, -self.clip_obs, self.clip_obs)
        acs = np.asarray(acs)
        if self.is_discrete:
            idx = acs.astype(int)
            if acs.ndim > 1:
                idx = np.squeeze(idx, axis=-1)
            acs = np.zeros([acs.shape[0], self.acs_dim])
            acs[np.arange(idx.shape[0]), idx] = 1.0
        if self.action_high is not None and self.action_low is not None:
            acs = np.clip(acs, self.action_low, self.action_high)
        x = np.concatenate([obs, acs], axis=-1)[..., self.select_dim]
        return th.tensor(x, dtype=th.float32)

    def forward(self, x):
        for i in range(self.n_layers):
            x = F.linear(x, self.params[f"{2 * i}.weight"], self.params[f"{2 * i}.bias"])
            if i < self.n_layers - 1:
                x = th.relu(x)
        return th.sigmoid(x)

    def cost_function(self, obs, acs):
        with th.no_grad():
            out = self.forward(self.prepare(obs, acs))
        return (1 - out.numpy()).squeeze(axis=-1)
