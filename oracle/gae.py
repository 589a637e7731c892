"""Oracle: dual (reward + cost) GAE over a [T, N] rollout.  Test infrastructure only.

ref: stable_baselines3/common/buffers.py:493-552
     (RolloutBufferWithCost._compute_returns_and_advantage / compute_returns_and_advantage)

The reference is numpy code whose dtypes matter:

  * ``dones`` stored in the buffer are float32, so for t < T-1
    ``next_non_terminal`` and ``delta`` are computed entirely in float32;
  * the final ``dones`` vector handed to ``compute_returns_and_advantage`` is a
    *bool* array, so ``1.0 - last_dones`` is float64 and the running
    ``last_gae_lam`` is float64 from the first (t = T-1) iteration onwards;
  * ``gamma * gae_lambda`` is a Python double that multiplies a float32 array
    for t < T-1 (rounded to float32 first) but a float64 array at t = T-1
    (where it multiplies the initial 0 and so never matters);
  * ``advantages[t] = last_gae_lam`` rounds the float64 accumulator to float32
    on store; ``returns = advantages + values`` is a float32 add.

Every step below spells those roundings out with explicit dtypes so the result
does not depend on the numpy version's promotion rules.
"""
import numpy as np

F32 = np.float32
F64 = np.float64


def gae_scan(rewards, values, dones, last_value, last_dones, gamma, gae_lambda):
    """One GAE pass.  rewards/values/dones: [T, N] float32 (dones[t] = done flag
    *entering* step t); last_value: [N] float32; last_dones: [N] bool.
    Returns (returns, advantages), both [T, N] float32."""
    rewards = np.ascontiguousarray(rewards, dtype=F32)
    values = np.ascontiguousarray(values, dtype=F32)
    dones = np.ascontiguousarray(dones, dtype=F32)
    last_value = np.asarray(last_value, dtype=F32).reshape(-1)
    last_dones = np.asarray(last_dones).astype(bool).reshape(-1)
    T = rewards.shape[0]
    g32 = F32(gamma)
    gl32 = F32(float(gamma) * float(gae_lambda))
    adv = np.empty_like(rewards)
    acc = None
    for t in range(T - 1, -1, -1):
        if t == T - 1:
            nnt = F64(1.0) - last_dones.astype(F64)
            gv = (g32 * last_value).astype(F32)                     # f32 product
            delta = rewards[t].astype(F64) + gv.astype(F64) * nnt   # f64 from here on
            delta = delta - values[t].astype(F64)
            acc = delta                                             # + coeff * 0
        else:
            nnt = (F32(1.0) - dones[t + 1]).astype(F32)
            gv = ((g32 * values[t + 1]).astype(F32) * nnt).astype(F32)
            delta = ((rewards[t] + gv).astype(F32) - values[t]).astype(F32)
            coeff = (gl32 * nnt).astype(F32)
            acc = delta.astype(F64) + coeff.astype(F64) * acc
        adv[t] = acc.astype(F32)
    returns = (adv + values).astype(F32)
    return returns, adv


def dual_gae(rewards, costs, reward_values, cost_values, dones,
             last_reward_value, last_cost_value, last_dones,
             reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda):
    """ref: buffers.py:543-552.  Returns dict with the four [T, N] float32 arrays."""
    r_ret, r_adv = gae_scan(rewards, reward_values, dones, last_reward_value, last_dones,
                            reward_gamma, reward_gae_lambda)
    c_ret, c_adv = gae_scan(costs, cost_values, dones, last_cost_value, last_dones,
                            cost_gamma, cost_gae_lambda)
    return dict(reward_returns=r_ret, reward_advantages=r_adv,
                cost_returns=c_ret, cost_advantages=c_adv)
