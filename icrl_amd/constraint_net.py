"""ConstraintNet (zeta_theta) — same constructor / methods as the reference, arithmetic in libicrl_hip.so.

ref: icrl/constraint_net.py:14-402.  Parameters live in one flat float32 HBM buffer in the reference's state_dict order
(network.0.weight, network.0.bias, ...).  Kept quirks: select_dim appends range(acs_dim) for the
action part (so it re-selects leading *observation* columns); ConstraintNet.load passes constructor arguments one slot off.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .structs import CnHyperT, CostNetT, p


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
        if len(hs) > 4 or min(hs, default=1) < 1:
            raise NotImplementedError(f"icrl_amd ConstraintNet supports 0..4 hidden layers (1-2 of up to 64 units inside the fused rollout, the rest "
                                      f"through the per-step launches), got {hs}")
        # 64-row activation images of the update / cost kernels (csrc/cn_train.hip: make_cn_dims) must fit the 160 KB LDS
        lds_floats = 64 * (self.input_dims + 1) + 64 * sum(h + 1 for h in hs) + 2 * 64 * (max(hs, default=0) + 1) + 128
        if lds_floats > 160 * 1024 // 4:
            raise NotImplementedError(f"icrl_amd ConstraintNet: hidden layers {hs} on {self.input_dims} inputs need {4 * lds_floats} B of LDS for 64 rows "
                                      "(160 KB available; e.g. 2 x 128, 3 x 96 or 4 x 64 units fit)")
        # a hidden layer above 64 units or more than two of them: the one-wave-per-row cost kernels (inside the fused rollout) do not hold
        # it; cost_function and train() then run 64 rows per workgroup (csrc/cn_train.hip: cn_cost_rows_kernel; weights from device memory
        # when they do not fit next to the activations)
        self.wide = max(hs, default=0) > 64 or len(hs) > 2 or len(hs) == 0
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
        h = hs + [0] * (4 - len(hs))
        return CostNetT(self.obs_dim, self.acs_dim, self.input_dims, len(hs), h[0], h[1], h[2], h[3],
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

    # ---- training (ref: constraint_net.py:137-229) ------------------------------------------------------------------------
    def _update_learning_rate(self, current_progress_remaining):
        self.current_progress_remaining = current_progress_remaining
        self.lr = float(self.lr_schedule(current_progress_remaining))

    def prepare_data(self, obs, acs):
        """ref: constraint_net.py:258-270 -> device float32 [N, input_dims]."""
        dev = self.device
        obs = torch.as_tensor(np.asarray(obs) if not torch.is_tensor(obs) else obs, device=dev).to(torch.float64).reshape(-1, self.obs_dim).contiguous()
        a_w = 1 if self.is_discrete else self.acs_dim
        acs = torch.as_tensor(np.asarray(acs) if not torch.is_tensor(acs) else acs, device=dev).to(torch.float32).reshape(-1, a_w).contiguous()
        out = torch.empty(obs.shape[0], self.input_dims, device=dev)
        s = self.struct()
        _lib.check(_lib.lib().icrl_cn_prepare(_lib.byref(s), p(obs), p(acs), obs.shape[0], p(out), _lib.current_stream()), "icrl_cn_prepare")
        return out

    def train(self, iterations, nominal_obs, nominal_acs, episode_lengths, obs_mean=None, obs_var=None,
              current_progress_remaining=1, perms=None):
        """ref: constraint_net.py:137-229.  `perms` ([iterations, min(Nn, Ne)], minibatch mode only) replaces the
        np.random.permutation draws of get() (:300-316); by default they are drawn from numpy's global generator, which is
        afterwards left where the reference would have left it (it stops drawing once an iteration early-stops)."""
        job = self._train_begin(iterations, nominal_obs, nominal_acs, episode_lengths, obs_mean, obs_var, current_progress_remaining, perms)
        self._train_launch(job)
        return self._train_end(job)

    # train() in three pieces (several runs sharing a GPU put their launches into one grid: icrl_amd/seed_batch.py)
    def _train_begin(self, iterations, nominal_obs, nominal_acs, episode_lengths, obs_mean=None, obs_var=None,
                     current_progress_remaining=1, perms=None):
        self._update_learning_rate(current_progress_remaining)
        self.current_obs_mean, self.current_obs_var = obs_mean, obs_var
        self._refresh_consts()
        nominal = self.prepare_data(nominal_obs, nominal_acs)
        # the reference re-runs prepare_data(expert) on every train() (constraint_net.py:157): the normalisation / clipping
        # constants may have changed since the last call.  Only the raw expert rows are cached on the device.
        if getattr(self, "_expert_dev", None) is None:
            a_w = 1 if self.is_discrete else self.acs_dim
            self._expert_dev = (torch.as_tensor(np.asarray(self.expert_obs), device=self.device).to(torch.float64).reshape(-1, self.obs_dim).contiguous(),
                                torch.as_tensor(np.asarray(self.expert_acs), device=self.device).to(torch.float32).reshape(-1, a_w).contiguous())
        expert = self.prepare_data(*self._expert_dev)
        dev, iters = self.device, int(iterations)
        lengths = np.asarray(episode_lengths, np.int64)
        offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
        assert offs[-1] == nominal.shape[0], "episode_lengths must cover the nominal rows"
        d_off = torch.as_tensor(offs, device=dev)
        d_rowep = torch.as_tensor(np.repeat(np.arange(len(lengths), dtype=np.int32), lengths), device=dev)
        L = _lib.lib()
        n_work = L.icrl_cn_train_work_floats(self.n_params, nominal.shape[0], expert.shape[0], len(lengths))
        work = torch.empty(n_work, device=dev)
        metrics = torch.zeros(max(iters, 1), 24, device=dev)
        t_dev = torch.tensor([self.adam_step], dtype=torch.int32, device=dev)
        hp = CnHyperT(iters, int(self.importance_sampling), int(self.per_step_importance_sampling), int(bool(self.train_gail_lambda)),
                      float(self.regularizer_coeff), float(self.eps), float(self.target_kl_old_new), float(self.target_kl_new_old),
                      float(self.lr), 0.9, 0.999, float(self.optimizer_kwargs.get("eps", 1e-8)))
        job = dict(s=self.struct(), hp=hp, nominal=nominal, expert=expert, d_off=d_off, d_rowep=d_rowep, n_ep=len(lengths), work=work,
                   metrics=metrics, t_dev=t_dev, iters=iters, rng_state=None, d_perms=None)
        if self.batch_size is not None:
            size = min(nominal.shape[0], expert.shape[0])
            if perms is None:
                job["rng_state"] = np.random.get_state()
                perms = np.stack([np.random.permutation(size) for _ in range(max(iters, 1))])
            if torch.is_tensor(perms):
                job["d_perms"] = perms[:max(iters, 1)].to(device=dev, dtype=torch.int32).contiguous()
            else:
                job["d_perms"] = torch.as_tensor(np.asarray(perms)[:max(iters, 1)].astype(np.int32), device=dev).contiguous()
            assert job["d_perms"].shape == (max(iters, 1), size), "perms must be [iterations, min(n_nominal, n_expert)]"
        return job

    def _train_launch(self, job):
        L, nominal, expert = _lib.lib(), job["nominal"], job["expert"]
        if self.batch_size is None:
            _lib.check(L.icrl_cn_train(_lib.byref(job["s"]), p(self.exp_avg), p(self.exp_avg_sq), p(job["t_dev"]), p(nominal), p(expert),
                                       nominal.shape[0], expert.shape[0], p(job["d_off"]), p(job["d_rowep"]), job["n_ep"], _lib.byref(job["hp"]),
                                       p(job["work"]), p(job["metrics"]), _lib.current_stream()), "icrl_cn_train")
        else:
            _lib.check(L.icrl_cn_train_minibatch(_lib.byref(job["s"]), p(self.exp_avg), p(self.exp_avg_sq), p(job["t_dev"]), p(nominal), p(expert),
                                                 nominal.shape[0], expert.shape[0], p(job["d_off"]), p(job["d_rowep"]), job["n_ep"],
                                                 _lib.byref(job["hp"]), p(job["d_perms"]), int(self.batch_size), p(job["work"]), p(job["metrics"]),
                                                 _lib.current_stream()), "icrl_cn_train_minibatch")

    def _train_end(self, job, metrics_host=None, adam_step_host=None):
        iters, nominal, expert, rng_state = job["iters"], job["nominal"], job["expert"], job["rng_state"]
        self.prepare()
        m = job["metrics"].cpu().numpy() if metrics_host is None else np.asarray(metrics_host, np.float32)          # the only host sync of the backward step
        self.adam_step = int(job["t_dev"].item()) if adam_step_host is None else int(adam_step_host)
        stopped = np.nonzero(m[:iters, 0] != 0)[0]
        early_stop_itr = int(stopped[0]) if len(stopped) else iters
        if rng_state is not None and early_stop_itr < iters:      # rewind to the reference's consumption of the stream
            np.random.set_state(rng_state)
            for _ in range(early_stop_itr):
                np.random.permutation(min(nominal.shape[0], expert.shape[0]))
        last_exec = early_stop_itr - 1 if len(stopped) else iters - 1       # iteration whose loss / preds the reference reports
        last_is = min(early_stop_itr, iters - 1)
        nanrow = np.full(24, np.nan, np.float32); nanrow[6] = np.inf
        lo = m[last_exec] if last_exec >= 0 else nanrow
        bw = {"backward/cn_loss": float(lo[6]), "backward/expert_loss": float(lo[7]),
              "backward/unweighted_nominal_loss": float(lo[8]), "backward/nominal_loss": float(lo[9]),
              "backward/regularizer_loss": float(lo[10]),
              "backward/is_mean": float(m[last_is][3]), "backward/is_max": float(m[last_is][4]), "backward/is_min": float(m[last_is][5]),
              "backward/nominal_preds_max": float(lo[11]), "backward/nominal_preds_min": float(lo[12]),
              "backward/nominal_preds_mean": float(lo[13]), "backward/expert_preds_max": float(lo[14]),
              "backward/expert_preds_min": float(lo[15]), "backward/expert_preds_mean": float(lo[16])}
        if self.importance_sampling:
            bw.update({"backward/kl_old_new": float(m[last_is][1]), "backward/kl_new_old": float(m[last_is][2]),
                       "backward/early_stop_itr": early_stop_itr})
        return bw

    # ---- persistence (ref: constraint_net.py:323-402) -----------------------------------------------------------------------
    def save(self, save_path):
        torch.save(dict(cn_network=self.state_dict(), cn_optimizer=dict(exp_avg=self.exp_avg.cpu(), exp_avg_sq=self.exp_avg_sq.cpu(),
                                                                         step=self.adam_step),
                        obs_dim=self.obs_dim, acs_dim=self.acs_dim, is_discrete=self.is_discrete, obs_select_dim=self.obs_select_dim,
                        acs_select_dim=self.acs_select_dim, clip_obs=self.clip_obs, obs_mean=self.current_obs_mean,
                        obs_var=self.current_obs_var, action_low=self.action_low, action_high=self.action_high,
                        device=str(self.device), hidden_sizes=self.hidden_sizes), save_path)

    @classmethod
    def load(cls, load_path, obs_dim=None, acs_dim=None, is_discrete=None, obs_select_dim=None, acs_select_dim=None,
             clip_obs=None, obs_mean=None, obs_var=None, action_low=None, action_high=None, device="auto"):
        """Loads the reference's .pt format.  The reference passes the constructor arguments positionally ONE SLOT OFF
        (constraint_net.py:394-399), so a loaded net has clip_obs=None, no action clipping, no observation normalisation and
        no optimizer — reproduced here because it is what the constraint-transfer runs (cpg) evaluate."""
        if isinstance(load_path, dict):
            sd = load_path
        elif str(load_path).endswith(".npz"):      # re-packed fixture of a reference checkpoint (tests/golden/cn_antbroken.npz)
            z = np.load(load_path)
            sd = dict(obs_dim=int(z["obs_dim"]), acs_dim=int(z["acs_dim"]), is_discrete=bool(z["is_discrete"]),
                      obs_select_dim=None, acs_select_dim=None, hidden_sizes=[int(h) for h in z["hidden_sizes"]],
                      cn_network={k[len("cn_network/"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith("cn_network/")})
        else:
            sd = torch.load(load_path, map_location="cpu", weights_only=False)
        g = lambda v, k: sd[k] if v is None else v
        obs_dim, acs_dim, is_discrete = g(obs_dim, "obs_dim"), g(acs_dim, "acs_dim"), g(is_discrete, "is_discrete")
        obs_select_dim, acs_select_dim = g(obs_select_dim, "obs_select_dim"), g(acs_select_dim, "acs_select_dim")
        net = cls(obs_dim, acs_dim, sd["hidden_sizes"], None, (lambda x: 0.0), None, None, is_discrete, 0.0,
                  obs_select_dim, acs_select_dim, optimizer_class=None, clip_obs=None, action_low=None, action_high=None)
        net.load_state_dict(sd["cn_network"])
        return net
