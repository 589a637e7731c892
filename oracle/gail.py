"""Oracle: the GAIL-constraint baseline's discriminator and its rollout-end hook.  Test infrastructure only.

ref: icrl/gail_utils.py:18-120   (GailDiscriminator: ReLU MLP + sigmoid over [obs, acs][select_dim]; NOTHING is normalised or
                                  clipped — normalize_obs is a no-op, clip_actions is commented out, :298-316)
     icrl/gail_utils.py:163-208  (train: per iteration one np.random.permutation(min(n_nominal, n_expert)) cut into batches
                                  (one batch of that size when batch_size is None: BOTH sets are truncated to it);
                                  loss = BCE(D(nominal), 0) + BCE(D(expert), 1); Adam eps 1e-5; 5 discriminator/* metrics)
     icrl/gail_utils.py:147-157  (reward_function: log(D + eps), or D itself)
     icrl/gail_utils.py:500-571  (GailCallback._on_rollout_end: the buffer's NORMALISED float32 observations are un-normalised
                                  with the statistics at the END of the rollout (not the raw observations), the discriminator
                                  takes one train() iteration on them, the buffer's rewards are replaced by (or, --learn_cost,
                                  incremented with) log D, returns / advantages are recomputed)
     icrl/gail.py:48-208         (plain PPO around it.  PPO.train (ppo/ppo.py:150-215) is PPOLagrangian.train without the cost
                                  term: the port runs its two-critics agent with zero costs, nu ~ 1e-8 and cost_vf_coef 0,
                                  which is the same arithmetic — checked bit for bit against the reference's PPO in g12)
"""
import numpy as np
import torch as th

from .gae import dual_gae
from .nets import CostNet


def make_disc(obs_dim, acs_dim, hidden, is_discrete=False, obs_select_dim=None, acs_select_dim=None, eps=1e-5):
    return CostNet(obs_dim, acs_dim, hidden, is_discrete, obs_select_dim, acs_select_dim, clip_obs=None, action_low=None,
                   action_high=None, eps=eps)


def flatten(x):
    x = np.asarray(x)
    if x.ndim > 2:
        d0, d1 = x.shape[:2]
        return x.reshape(d0 * d1, -1), (d0, d1)
    return x, (x.shape[0], 1)


def disc_train(net, optimizer, iterations, nominal_obs, nominal_acs, expert_obs, expert_acs, batch_size=None, rng=np.random,
               freeze=False):
    nominal = net.prepare(flatten(nominal_obs)[0], flatten(nominal_acs)[0])
    expert = net.prepare(expert_obs, expert_acs)
    bce = th.nn.BCELoss()
    size = min(nominal.shape[0], expert.shape[0])
    for _ in range(iterations):
        perm = rng.permutation(size)
        bs = size if batch_size is None else batch_size
        for s in range(0, size, bs):
            idx = perm[s:s + bs]
            nominal_preds, expert_preds = net.forward(nominal[idx]), net.forward(expert[idx])
            nominal_loss = bce(nominal_preds, th.zeros(*nominal_preds.size()))
            expert_loss = bce(expert_preds, th.ones(*expert_preds.size()))
            loss = nominal_loss + expert_loss
            if not freeze:
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
    return {"discriminator/disc_loss": loss.item(), "discriminator/expert_loss": expert_loss.item(),
            "discriminator/nominal_loss": nominal_loss.item(), "discriminator/mean_nominal_preds": nominal_preds.mean().item(),
            "discriminator/mean_expert_preds": expert_preds.mean().item()}


def disc_reward(net, obs, acs, apply_log=True):
    o, shape = flatten(obs)
    a, _ = flatten(acs)
    with th.no_grad():
        d = net.forward(net.prepare(o, a)).numpy().reshape(shape)
    return np.squeeze(np.log(d + net.eps)) if apply_log else np.squeeze(d)


def unnormalize_obs(norm, obs):
    """ref: vec_normalize.py:125-128 — with the CURRENT statistics."""
    if not norm.norm_obs:
        return obs
    return (obs * np.sqrt(norm.obs_rms.var + norm.epsilon)) + norm.obs_rms.mean


def rollout_end(agent, net, optimizer, expert_obs, expert_acs, last_v_r, last_dones, *, batch_size=None, learn_cost=False,
                true_cost_fn=None, rng=np.random, freeze=False):
    """GailCallback._on_rollout_end on a PortAgent whose buffer has just been collected."""
    buf = agent.buf
    obs = unnormalize_obs(agent.stack.norm, buf.observations.copy())
    acs = buf.actions.copy()
    m = disc_train(net, optimizer, 1, obs, acs, expert_obs, expert_acs, batch_size, rng, freeze)
    if true_cost_fn is not None:
        m["eval/mean_cost"] = float(np.mean(true_cost_fn(obs.reshape(-1, obs.shape[-1]), acs.reshape(-1, acs.shape[-1]))))
    rew = disc_reward(net, obs, acs)
    assert rew.shape == buf.rewards.shape
    buf.rewards = buf.rewards + rew if learn_cost else rew
    g = dual_gae(buf.rewards, buf.costs, buf.reward_values, buf.cost_values, buf.dones, last_v_r, np.zeros_like(last_v_r), last_dones,
                 *agent.gammas)
    buf.reward_returns, buf.reward_advantages = g["reward_returns"], g["reward_advantages"]
    return m
