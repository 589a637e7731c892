"""GAIL-constraint baseline: discriminator and rollout-end callback, device-resident.

ref: icrl/gail_utils.py:18-498 (GailDiscriminator), :500-571 (GailCallback).  The discriminator is the constraint net read the
other way round — the same ReLU MLP + sigmoid, trained with BCE on nominal (label 0) vs expert (label 1) rows — so it runs on
the constraint-net kernels (icrl_cn_prepare, icrl_cn_train_minibatch in its BCE mode, icrl_disc_reward).  Reference behaviour
kept on purpose: nothing is normalised or clipped on the way in (normalize_obs is a no-op and clip_actions is commented out,
gail_utils.py:298-316), both sets are truncated to min(n_nominal, n_expert) rows by the shared permutation of get(), and the
callback un-normalises the buffer's float32 NORMALISED observations with the statistics at the end of the rollout instead of
using the raw observations (gail_utils.py:539-542).
"""
import numpy as np
import torch

from . import _lib, callbacks, logger
from .constraint_net import ConstraintNet
from .structs import p


class GailDiscriminator(ConstraintNet):
    def __init__(self, obs_dim, acs_dim, hidden_sizes, batch_size, lr_schedule, expert_obs, expert_acs, is_discrete,
                 obs_select_dim=None, acs_select_dim=None, optimizer_class=torch.optim.Adam, optimizer_kwargs=None, clip_obs=10.,
                 initial_obs_mean=None, initial_obs_var=None, action_low=None, action_high=None, num_spurious_features=None,
                 freeze_weights=False, eps=1e-5, device="cuda"):
        if num_spurious_features is not None:
            raise NotImplementedError("num_spurious_features (a diagnostic of the reference's grid-world study) is not built")
        super().__init__(obs_dim, acs_dim, hidden_sizes, batch_size, lr_schedule, expert_obs, expert_acs, is_discrete, 0.0,
                         obs_select_dim, acs_select_dim, optimizer_class, optimizer_kwargs, no_importance_sampling=True,
                         clip_obs=None, action_low=None, action_high=None, train_gail_lambda=True, eps=eps, device=device)
        # stored like the reference stores them; its prepare_*_data never applies them
        self.stored_clip_obs, self.stored_action_low, self.stored_action_high = clip_obs, action_low, action_high
        self.freeze_weights = freeze_weights

    @staticmethod
    def flatten(x):
        x = torch.as_tensor(x) if not torch.is_tensor(x) else x
        if x.dim() > 2:
            d0, d1 = x.shape[:2]
            return x.reshape(d0 * d1, -1), (d0, d1)
        return x, (x.shape[0], 1)

    def train(self, iterations, nominal_obs, nominal_acs, obs_mean=None, obs_var=None, current_progress_remaining=1, perms=None):
        """ref: gail_utils.py:163-208 -> the five discriminator/* metrics of the last minibatch."""
        obs, _ = self.flatten(nominal_obs)
        acs, _ = self.flatten(nominal_acs)
        n_exp = int(np.asarray(self.expert_obs).shape[0])
        size = min(int(obs.shape[0]), n_exp)
        keep_bs, keep_sched = self.batch_size, self.lr_schedule
        self.batch_size = size if keep_bs is None else int(keep_bs)     # batch_size None: ONE batch of `size` permuted rows
        if self.freeze_weights:
            self.lr_schedule = lambda _x: 0.0                            # evaluated, not updated
        try:
            m = super().train(int(iterations), obs, acs, np.array([obs.shape[0]]), None, None, current_progress_remaining, perms=perms)
        finally:
            self.batch_size, self.lr_schedule = keep_bs, keep_sched
        return {"discriminator/disc_loss": m["backward/cn_loss"], "discriminator/expert_loss": m["backward/expert_loss"],
                "discriminator/nominal_loss": m["backward/nominal_loss"],
                "discriminator/mean_nominal_preds": m["backward/nominal_preds_mean"],
                "discriminator/mean_expert_preds": m["backward/expert_preds_mean"]}

    def reward_function(self, obs, acs, apply_log=True):
        """ref: gail_utils.py:147-157 -> device float32 tensor shaped like the leading axes of `obs`."""
        o, shape = self.flatten(obs)
        a, _ = self.flatten(acs)
        assert o.shape[-1] == self.obs_dim, ""
        dev = self.device
        o = o.to(device=dev, dtype=torch.float64).contiguous()
        a = a.to(device=dev, dtype=torch.float32).reshape(o.shape[0], -1).contiguous()
        out = torch.empty(o.shape[0], device=dev)
        s = self.struct()
        _lib.check(_lib.lib().icrl_disc_reward(_lib.byref(s), p(o), p(a), o.shape[0], p(out), int(bool(apply_log)),
                                               _lib.current_stream()), "icrl_disc_reward")
        return out.reshape(shape).squeeze()

    def save(self, save_path):
        torch.save(dict(network=self.state_dict(), optimizer=dict(exp_avg=self.exp_avg.cpu(), exp_avg_sq=self.exp_avg_sq.cpu(), step=self.adam_step),
                        obs_dim=self.obs_dim, acs_dim=self.acs_dim, is_discrete=self.is_discrete, obs_select_dim=self.obs_select_dim,
                        acs_select_dim=self.acs_select_dim, clip_obs=self.stored_clip_obs, obs_mean=self.current_obs_mean,
                        obs_var=self.current_obs_var, action_low=self.stored_action_low, action_high=self.stored_action_high,
                        device=str(self.device), hidden_sizes=self.hidden_sizes), save_path)

    @classmethod
    def load(cls, load_path, obs_dim=None, acs_dim=None, is_discrete=None, expert_obs=None, expert_acs=None, obs_select_dim=None,
             acs_select_dim=None, clip_obs=None, obs_mean=None, obs_var=None, action_low=None, action_high=None, device="auto"):
        """ref: gail_utils.py:356-402 (reads the reference's gail_discriminator.pt: key `network`)."""
        sd = load_path if isinstance(load_path, dict) else torch.load(load_path, map_location="cpu", weights_only=False)
        g = lambda v, k: sd[k] if v is None else v
        net = cls(g(obs_dim, "obs_dim"), g(acs_dim, "acs_dim"), sd["hidden_sizes"], None, (lambda x: 0.0), expert_obs, expert_acs,
                  g(is_discrete, "is_discrete"), g(obs_select_dim, "obs_select_dim"), g(acs_select_dim, "acs_select_dim"), None, None,
                  g(clip_obs, "clip_obs"), g(obs_mean, "obs_mean"), g(obs_var, "obs_var"), g(action_low, "action_low"),
                  g(action_high, "action_high"))
        net.load_state_dict(sd["network"])
        return net


class GailCallback(callbacks.BaseCallback):
    """ref: gail_utils.py:500-571 — at the end of every rollout, before the policy update: one discriminator iteration on the
    rollout, eval/mean_cost on the true cost, the buffer's rewards relabelled with log D, returns and advantages recomputed."""

    def __init__(self, discriminator, learn_cost, true_cost_function, save_dir=None, plot_disc=False, update_freq=1, verbose=1):
        super().__init__(verbose)
        self.discriminator, self.update_freq, self.learn_cost = discriminator, update_freq, learn_cost
        self.true_cost_function = true_cost_function
        self.disc_itr, self.history, self.perms = 0, [], None

    def _on_rollout_end(self):
        rb, env, model = self.model.rollout_buffer, self.training_env, self.model
        obs = rb.observations.double()
        if getattr(env, "norm_obs", False):          # VecNormalize.unnormalize_obs with the CURRENT statistics (vec_normalize.py:125-128)
            obs = obs * torch.sqrt(env.obs_rms.d_var + env.epsilon) + env.obs_rms.d_mean
        acs = rb.actions
        rec = {}
        if self.disc_itr % self.update_freq == 0:
            self.discriminator.current_obs_mean, self.discriminator.current_obs_var = env.obs_rms.mean, env.obs_rms.var
            rec = self.discriminator.train(1, obs, acs, perms=None if self.perms is None else self.perms(self.disc_itr))
            for k, v in rec.items():
                logger.record(k, v)
        c = self.true_cost_function(obs.reshape(-1, obs.shape[-1]), acs.reshape(-1, acs.shape[-1]))
        rec["eval/mean_cost"] = float(c.double().mean().item()) if torch.is_tensor(c) else float(np.mean(c))
        logger.record("eval/mean_cost", rec["eval/mean_cost"])
        rewards = self.discriminator.reward_function(obs, acs)
        assert rewards.shape == rb.rewards.shape
        if self.learn_cost:
            rb.rewards += rewards
        else:
            rb.rewards.copy_(rewards)
        rb.compute_returns_and_advantage(model._ag["last_v_r"], model._ag["last_v_c"], model._ag["last_dones"])
        self.history.append(rec)
        self.disc_itr += 1
