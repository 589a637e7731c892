"""Oracle: PPO-Lagrangian minibatch update, epoch loop and the dual (Lagrange multiplier) step.
Test infrastructure only.

ref: stable_baselines3/ppo_lag/ppo_lag.py:177-338       (PPOLagrangian.train)
     stable_baselines3/common/buffers.py:594-627         (RolloutBufferWithCost.get / _get_samples)
     stable_baselines3/common/buffers.py:53-65           (swap_and_flatten: [T,N,..] -> env-major [N*T,..])
     stable_baselines3/common/dual_variable.py:9-57      (Nu, DualVariable)

Third-party arithmetic (torch 2.10 here, 1.5 in the reference's README): Adam, clip_grad_norm_,
softplus, autograd.  ``adam_step_explicit`` / ``clip_coef_explicit`` restate the published
formulas the HIP kernels implement, and tests check them against torch's own.
"""
import math

import numpy as np
import torch as th
import torch.nn.functional as F


def env_major(x):
    """[T, N, ...] -> [N*T, ...] with flat index i = env * T + t (ref: buffers.py:53-65)."""
    x = np.asarray(x)
    if x.ndim < 3:
        x = x[..., None]
    return x.swapaxes(0, 1).reshape(x.shape[0] * x.shape[1], *x.shape[2:])


def minibatch_loss(policy, obs, actions, old_log_prob, adv_r, adv_c, ret_r, ret_c, old_v_r, old_v_c, nu,
                   clip_range, ent_coef=0.0, reward_vf_coef=0.5, cost_vf_coef=0.5,
                   clip_range_reward_vf=None, clip_range_cost_vf=None):
    """ref: ppo_lag.py:201-281.  All tensor args are torch fp32 minibatch slices; ``nu`` is the detached
    Python float the reference reads with .item().  Returns (loss, terms dict of tensors)."""
    v_r, v_c, log_prob, entropy = policy.evaluate_actions(obs, actions)
    v_r, v_c = v_r.flatten(), v_c.flatten()
    a_r = adv_r - adv_r.mean()
    a_r = a_r / (adv_r.std() + 1e-8)                      # torch .std() is the unbiased estimator
    a_c = adv_c - adv_c.mean()                            # centred, NOT rescaled (ref: :222)
    ratio = th.exp(log_prob - old_log_prob)
    pl1 = a_r * ratio
    pl2 = a_r * th.clamp(ratio, 1 - clip_range, 1 + clip_range)
    policy_loss = -th.min(pl1, pl2).mean()
    policy_loss = policy_loss + nu * th.mean(a_c * ratio)  # cost term is unclipped (ref: :234-235)
    policy_loss = policy_loss / (1 + nu)
    clip_fraction = th.mean((th.abs(ratio - 1) > clip_range).float())
    v_r_pred = v_r if clip_range_reward_vf is None else \
        old_v_r + th.clamp(v_r - old_v_r, -clip_range_reward_vf, clip_range_reward_vf)
    v_c_pred = v_c if clip_range_cost_vf is None else \
        old_v_c + th.clamp(v_c - old_v_c, -clip_range_cost_vf, clip_range_cost_vf)
    reward_value_loss = F.mse_loss(ret_r, v_r_pred)
    cost_value_loss = F.mse_loss(ret_c, v_c_pred)
    entropy_loss = -th.mean(-log_prob) if entropy is None else -th.mean(entropy)
    loss = policy_loss + ent_coef * entropy_loss + reward_vf_coef * reward_value_loss + cost_vf_coef * cost_value_loss
    approx_kl = th.mean(old_log_prob - log_prob).detach()
    return loss, dict(policy_loss=policy_loss.detach(), reward_value_loss=reward_value_loss.detach(),
                      cost_value_loss=cost_value_loss.detach(), entropy_loss=entropy_loss.detach(),
                      clip_fraction=clip_fraction, approx_kl=approx_kl, log_prob=log_prob.detach(),
                      v_r=v_r.detach(), v_c=v_c.detach())


def ppo_lag_train(policy, optimizer, buf, perms, nu, *, batch_size, n_epochs, clip_range, target_kl=None,
                  max_grad_norm=0.5, ent_coef=0.0, reward_vf_coef=0.5, cost_vf_coef=0.5, discrete=False,
                  clip_range_reward_vf=None, clip_range_cost_vf=None, max_minibatches=None):
    """The epoch loop of ``train()`` (ref: ppo_lag.py:196-299).

    buf:   dict of [T, N(, d)] float32 arrays (observations, actions, log_probs, reward_values,
           reward_advantages, reward_returns, cost_values, cost_advantages, cost_returns).
    perms: callable epoch -> permutation of T*N (the reference draws np.random.permutation per epoch),
           or an [n_epochs, T*N] integer array (teacher-forced).
    Returns dict of the train/* scalars that depend on the loop."""
    flat = {k: th.as_tensor(env_major(buf[k])) for k in
            ("observations", "actions", "log_probs", "reward_values", "reward_advantages", "reward_returns",
             "cost_values", "cost_advantages", "cost_returns")}
    n = flat["observations"].shape[0]
    bs = n if batch_size is None else batch_size
    ent, pg, rvl, cvl, cfr, all_kl = [], [], [], [], [], []
    early_stop_epoch = n_epochs
    loss = None
    kls = []
    for epoch in range(n_epochs):
        idx = perms(epoch) if callable(perms) else np.asarray(perms[epoch])
        kls = []
        n_mb = 0
        for start in range(0, n, bs):
            b = th.as_tensor(np.asarray(idx[start:start + bs]), dtype=th.long)
            actions = flat["actions"][b]
            if discrete:
                actions = actions.long().flatten()
            loss, tr = minibatch_loss(
                policy, flat["observations"][b], actions, flat["log_probs"][b].flatten(),
                flat["reward_advantages"][b].flatten(), flat["cost_advantages"][b].flatten(),
                flat["reward_returns"][b].flatten(), flat["cost_returns"][b].flatten(),
                flat["reward_values"][b].flatten(), flat["cost_values"][b].flatten(), nu, clip_range,
                ent_coef, reward_vf_coef, cost_vf_coef, clip_range_reward_vf, clip_range_cost_vf)
            optimizer.zero_grad()
            loss.backward()
            th.nn.utils.clip_grad_norm_(policy.parameters(), max_grad_norm)
            optimizer.step()
            pg.append(tr["policy_loss"].item()); cfr.append(tr["clip_fraction"].item())
            rvl.append(tr["reward_value_loss"].item()); cvl.append(tr["cost_value_loss"].item())
            ent.append(tr["entropy_loss"].item()); kls.append(tr["approx_kl"].cpu().numpy())
            n_mb += 1
            if max_minibatches is not None and n_mb >= max_minibatches:
                break
        all_kl.append(np.mean(kls))
        if target_kl is not None and np.mean(kls) > 1.5 * target_kl:
            early_stop_epoch = epoch
            break
    return {"train/entropy_loss": np.mean(ent), "train/policy_gradient_loss": np.mean(pg),
            "train/reward_value_loss": np.mean(rvl), "train/cost_value_loss": np.mean(cvl),
            "train/approx_kl": np.mean(kls), "train/clip_fraction": np.mean(cfr),
            "train/loss": loss.item(), "train/early_stop_epoch": early_stop_epoch,
            "epoch_kls": np.asarray(all_kl, dtype=np.float64)}


# ---------------------------------------------------------------------------------------------
# explicit restatements of the third-party optimiser arithmetic used by the HIP kernels
# ---------------------------------------------------------------------------------------------

def clip_coef_explicit(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: total = ||(||g_1||, ..., ||g_k||)||_2; coef = min(1, max_norm/(total+1e-6))."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = max_norm / (total + 1e-6)
    return total, min(coef, 1.0)


def adam_step_explicit(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay / amsgrad), single-tensor form:
       m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ---------------------------------------------------------------------------------------------
# dual variable
# ---------------------------------------------------------------------------------------------

def inv_softplus_floor(x):
    """log(max(exp(x) - 1, 1e-8)) (ref: dual_variable.py:19,28-29)."""
    return float(np.log(max(np.exp(x) - 1, 1e-8)))


class Dual:
    """nu = softplus(log_nu) trained with Adam on loss = -nu * (cost - alpha), then clamped from below.
    ref: dual_variable.py:9-57.  Quirk kept: ``clamp_at`` defaults to the *already inverse-softplused*
    init, and clamp() applies the inverse softplus again."""

    def __init__(self, alpha=0.0, learning_rate=10.0, penalty_init=1.0, clamp_at=None):
        init = inv_softplus_floor(penalty_init)
        self.log_nu = (init * th.ones(1)).requires_grad_(True)
        self.clamp_at = init if clamp_at is None else clamp_at
        self.alpha = alpha
        self.opt = th.optim.Adam([self.log_nu], lr=learning_rate)
        self.loss = th.tensor(0)

    def nu(self):
        return F.softplus(self.log_nu)

    def update(self, cost):
        self.loss = -self.nu() * (cost - self.alpha)
        self.opt.zero_grad()
        self.loss.backward()
        self.opt.step()
        self.log_nu.data.clamp_(min=inv_softplus_floor(self.clamp_at))


class PID:
    """PID controller on the Lagrange multiplier (cpg --use_pid).  ref: dual_variable.py:60-122: integral term clipped at 0,
    proportional term on an EMA of (cost - budget), derivative term max(0, EMA(cost) - EMA(cost) pid_delay updates ago);
    nu() hands the penalty out as a float32 tensor."""

    def __init__(self, alpha=0, penalty_init=1, Kp=0, Kd=0, Ki=1, pid_delay=10, delta_d_ema_alpha=0.95, delta_p_ema_alpha=0.95):
        from collections import deque
        self.budget, self.Kp, self.Ki, self.Kd = alpha, Kp, Ki, Kd
        self.pid_i = self.cost_penalty = penalty_init
        self.history = deque([0], maxlen=pid_delay)
        self.delta_p = self.cost_ema = 0
        self.a_d, self.a_p = delta_d_ema_alpha, delta_p_ema_alpha
        self.loss = th.tensor(0.0)

    def update(self, cost):
        cost = float(cost)
        self.loss = th.tensor(cost)
        delta = cost - self.budget
        self.pid_i = max(0, self.pid_i + self.Ki * delta)
        self.delta_p = self.a_p * self.delta_p + (1 - self.a_p) * delta
        self.cost_ema = self.a_d * self.cost_ema + (1 - self.a_d) * cost
        pid_d = max(0, self.cost_ema - self.history[0])
        self.cost_penalty = max(0, self.Kp * self.delta_p + self.Kd * pid_d + self.pid_i)
        self.history.append(self.cost_ema)

    def nu(self):
        return th.tensor(self.cost_penalty)
