"""GPU: parity at the size the bench quotes (BASELINE configs[1]) and at configs[2]'s widths — long dependent chains.

The other oracle tests stop at <= 960 dependent optimiser steps; bench.py's `value` is measured on 20 480 dependent steps per
train() (64 envs x 2048 rows, 10 epochs x 2048 minibatches of 64).  Here the HIP path and the CPU port (oracle.loop.PortAgent,
pinned to the reference by g4 / g9 / g13) run ONE forward step of an outer iteration at exactly that size on the same
SeededStreams — 2 rollouts + 2 train() calls, README.md:38 flags — and the drift of everything the reference logs is measured
after 20 480 and 40 960 dependent steps (ref: stable_baselines3/ppo_lag/ppo_lag.py:196-299, common/buffers.py:493-541).

What holds (asserted) and what was measured on MI355X (printed by the test, recorded in DESIGN.md section 2):
  * the target-KL early-stop decision of every epoch is the same on both sides (`early_stop_epoch` equal); the per-epoch mean
    approx-KL values differ by fp32 rounding order only and are not within 2 % of the 1.5 x target_kl threshold when a decision
    is taken, so the decision is not at the mercy of summation order here; if an epoch ever sits that close the test says so
    instead of failing on a coin flip;
  * nu (the Lagrange multiplier trajectory) within 1e-5 absolute after each train();
  * parameters within ADAM_DEV_BOUND x lr x steps of the port's (a drift bound, see below: over 2 x 10^4 dependent steps the fp32
    trajectories separate faster than the per-step rate of the short tests).
"""
import os
import time

import numpy as np
import pytest
import torch

from oracle import loop as o_loop, nets as o_nets
from oracle.streams import SeededStreams

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

# CALIBRATED bounds (rounds 4 / 5).  The short tests (<= 48 dependent steps, tests/test_ppo_train_gpu.py) hold |dp| <= 5e-4 x lr x steps; over
# 2 x 10^4 dependent steps two fp32 executions of the SAME algorithm separate faster than that (Adam turns a rounding-size difference
# in a near-zero gradient into an lr-size difference in the update).  How fast is measured, not fitted: tools/calibrate_drift.py runs the
# CPU port against ITSELF through this file's two schedules with a rounding-size disturbance — every initial parameter moved by one
# float32 ulp up / down / at random, the rows of every minibatch reversed or rotated (another summation order in every step), 8 torch
# threads instead of 1 — profiles/r05_drift_calibration.md (15 disturbances for configs[1], 21 for configs[2]):
#                                port vs disturbed port, configs[1] (15 runs)                         HIP vs port (MI355X, rounds 4-5)
#   max |dp| after 20 480 steps  1.05-2.64e-3 x lr x steps (median 1.23e-3, 90 % 1.50e-3)             1.1e-3 x lr x steps
#   max |dp| after 40 960 steps  1.05-1.82e-3 x lr x steps (median 1.29e-3, 90 % 1.45e-3)             1.0-1.8e-3: the upper end of the distribution, not above it
#   nu                           0 | <= 8.3e-7                                                          0 | 3.6e-7
#   average_cost                 0 | <= 5.6e-4 (rollout 2 is collected with the drifted parameters)     1.2e-7 | 1.0-3.4e-4
#   losses pg / rv / cv          <= 1.9e-6 | <= 3.0e-5 / 1.2e-4 / 9.6e-5                                <= 6.9e-7 | 7.9e-6 / 2.6e-5 / 6.4e-5
#   early_stop_epoch             equal                                                                  equal
# Each bound below is 2 x the largest port-vs-port figure (first | second train()).
ADAM_DEV_BOUND = (5.3e-3, 3.7e-3)            # x lr x steps: 2 x 2.64e-3 | 2 x 1.82e-3
NU_BOUND = 1.7e-6                            # 2 x 8.3e-7
AVERAGE_COST_BOUND = (1e-6, 1.2e-3)          # (0 measured: the floor of one float32 mean) | 2 x 5.6e-4
LOSS_BOUND = (4e-6, 2.4e-4)                  # 2 x 1.9e-6 | 2 x 1.2e-4
# configs[2]'s schedule (lr 3e-5, 3 072 + 1 280 executed steps) amplifies nothing: 21 disturbed runs of the port stay within 1-3 ulp of
# the undisturbed one — max |dp| <= 3.2e-6 x lr x steps (3e-7 absolute), nu 0, average_cost <= 3e-8, losses <= 1.4e-8.  HIP vs port on
# MI355X: 3.5e-6 / 4.1e-6 x lr x steps, nu 7.5e-9 (ONE float32 ulp of 0.1), average_cost 3e-8.  Bounds: parameters 2 x the port-vs-port
# maximum; nu and average_cost two float32 ulps of their values (port-vs-port is 0 / one ulp); the LOSSES are not drift-bounded at all
# here — the port's disturbed runs share libm's tanh / exp, the HIP kernels' v_exp-based ones differ by ~1e-6 relative in every
# evaluation (rv 7.3e-7, cv 4.9e-7 measured) — they get half the kernel-level tolerance north_star states (1e-5 relative).
ANT_DEV_BOUND = 6.5e-6                       # x lr x steps: 2 x 3.23e-6
ANT_NU_BOUND = 1.5e-8                        # two float32 ulps at nu ~ 0.1
ANT_AVERAGE_COST_BOUND = 1.2e-7              # two float32 ulps at ~ 0.5 (port-vs-port: 3e-8)
ANT_LOSS_RTOL = 5e-6                         # + 2e-8 absolute (2 x the port-vs-port maximum)


def _pair(env_id, kind, N, T, od, ad, cn_layers, seed, **kw):
    from icrl_amd import utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    env = utils.make_train_env(env_id, None, True, seed, N, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    lo = -np.ones(ad, np.float32)
    torch.manual_seed(seed + 1)
    cn = ConstraintNet(od, ad, cn_layers, None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    ocn = o_nets.CostNet(od, ad, cn_layers, False, None, None, 20, lo, -lo)
    ocn.load_state_dict(cn.state_dict())
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=seed, streams=SeededStreams(77), **kw)
    stack = o_loop.make_stack(N, kind, seed); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, seed=seed, **kw)
    port.policy.load_state_dict(agent.policy.state_dict())
    return agent, port, env


@pytest.fixture(autouse=True)
def _one_cpu_thread():
    """the port's 64-row MLP steps are fastest on one thread (bench.py: cpu_baseline); restored afterwards — the thread count
    changes the rounding of torch's CPU initialisers (orthogonal_: LAPACK QR), which other tests compare across processes."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _forward_step(agent, port, env, n_rollouts, lr, target_kl, calibrated="configs1"):
    """learn() of both sides, rollout by rollout, with a comparison after every train()."""
    from icrl_amd import logger
    T, N = agent.n_steps, agent.n_envs
    streams = SeededStreams(77)
    agent._setup_learn(n_rollouts * N * T)
    port.num_timesteps = 0
    port._last_obs = port.stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = port.stack.old_obs.copy()
    rows, t_port = [], 0.0
    for k in range(n_rollouts):
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        agent.train()
        lg = dict(logger.Logger.CURRENT.name_to_value)
        t0 = time.time()
        b = port.collect_rollouts(streams.rollout_noise(T, N, port.act_dim))
        out = port.train(lambda e: streams.permutation(e, T * N))
        streams.consumed(min(int(out["train/early_stop_epoch"]) + 1, agent.n_epochs))
        t_port += time.time() - t0
        steps = agent.policy.adam_step
        worst_abs, worst_rate = 0.0, 0.0
        for name, v in agent.policy.state_dict().items():
            d = float(np.abs(v.numpy() - port.policy.params[name].detach().numpy()).max())
            worst_abs = max(worst_abs, d)
        worst_rate = worst_abs / (lr * steps)
        rb = agent.rollout_buffer
        buf_dev = {f: float(np.abs(getattr(rb, f).cpu().numpy().reshape(getattr(b, f).shape) - getattr(b, f)).max())
                   for f in ("rewards", "costs", "orig_costs", "log_probs", "reward_values", "reward_advantages", "cost_advantages")}
        kls_hip = np.asarray(agent.epoch_kls, np.float64)
        kls_port = np.asarray(out.get("epoch_kls", []), np.float64)
        rows.append(dict(rollout=k, steps=steps, nu=(lg["train/nu"], out["train/nu"]), average_cost=(lg["train/average_cost"], out["train/average_cost"]),
                         early_stop_epoch=(lg["train/early_stop_epoch"], out["train/early_stop_epoch"]),
                         pg_loss=(lg["train/policy_gradient_loss"], out["train/policy_gradient_loss"]),
                         rv_loss=(lg["train/reward_value_loss"], out["train/reward_value_loss"]),
                         cv_loss=(lg["train/cost_value_loss"], out["train/cost_value_loss"]),
                         approx_kl=(lg["train/approx_kl"], out["train/approx_kl"]),
                         clip_fraction=(lg["train/clip_fraction"], out["train/clip_fraction"]),
                         param_dev=worst_abs, param_rate=worst_rate, buffer_dev=buf_dev, kls_hip=kls_hip, kls_port=kls_port))
        r = rows[-1]
        print(f"[full-size] after train() #{k + 1}: {steps} dependent optimiser steps; nu {r['nu'][0]:.9f} vs {r['nu'][1]:.9f} (d {abs(r['nu'][0] - r['nu'][1]):.2e}); "
              f"average_cost d {abs(r['average_cost'][0] - r['average_cost'][1]):.2e}; early_stop_epoch {r['early_stop_epoch']}; "
              f"losses d pg {abs(r['pg_loss'][0] - r['pg_loss'][1]):.2e} rv {abs(r['rv_loss'][0] - r['rv_loss'][1]):.2e} cv {abs(r['cv_loss'][0] - r['cv_loss'][1]):.2e} "
              f"(values {r['pg_loss'][1]:.3g} / {r['rv_loss'][1]:.3g} / {r['cv_loss'][1]:.3g}); "
              f"approx_kl d {abs(r['approx_kl'][0] - r['approx_kl'][1]):.2e}; clip_fraction d {abs(r['clip_fraction'][0] - r['clip_fraction'][1]):.2e}; "
              f"max |d param| {worst_abs:.3e} = {worst_rate:.2e} x lr x steps; buffer of this rollout: "
              + ", ".join(f"{f} {v:.1e}" for f, v in buf_dev.items()))
        # ---- what must hold
        ee_h, ee_p = int(r["early_stop_epoch"][0]), int(r["early_stop_epoch"][1])
        if ee_h != ee_p and target_kl is not None:
            # a decision within fp32 rounding of the threshold would be a coin flip, not a bug: say which epoch and how close
            thr = 1.5 * target_kl
            e = min(ee_h, ee_p)
            close = abs(kls_hip[e] - thr) / thr if e < len(kls_hip) else float("nan")
            assert close < 2e-3, f"early stop differs (HIP {ee_h}, port {ee_p}) and epoch {e}'s mean KL {kls_hip[e]} is not at the threshold {thr}"
            # not a skip (invisible in a -q pass count): the run up to here was compared, the rest of it has legitimately diverged
            import warnings
            warnings.warn(f"full-size parity: epoch {e}'s mean approx-KL {kls_hip[e]:.7f} sits within {close:.1e} of the 1.5 x target_kl threshold; "
                          f"the early-stop decision flipped with summation order (HIP {ee_h}, port {ee_p}); comparison stops after train() #{k + 1}")
            return rows
        if calibrated == "configs2":
            assert abs(r["nu"][0] - r["nu"][1]) <= ANT_NU_BOUND, r["nu"]
            assert abs(r["average_cost"][0] - r["average_cost"][1]) <= ANT_AVERAGE_COST_BOUND, r["average_cost"]
            assert worst_abs <= ANT_DEV_BOUND * lr * steps + 1e-9, (worst_abs, steps)
            for key in ("pg_loss", "rv_loss", "cv_loss"):
                assert abs(r[key][0] - r[key][1]) <= 2e-8 + ANT_LOSS_RTOL * abs(r[key][1]), (key, r[key])
            continue
        assert abs(r["nu"][0] - r["nu"][1]) <= NU_BOUND, r["nu"]
        # (the SECOND rollout is collected with parameters that already differ by ~1e-2: its mean cost moves by a few 1e-4)
        assert abs(r["average_cost"][0] - r["average_cost"][1]) <= AVERAGE_COST_BOUND[min(k, 1)]
        assert worst_abs <= ADAM_DEV_BOUND[min(k, 1)] * lr * steps + 2e-7, (worst_abs, steps)
        for key in ("pg_loss", "rv_loss", "cv_loss"):
            assert abs(r[key][0] - r[key][1]) <= LOSS_BOUND[min(k, 1)], (key, r[key])
    print(f"[full-size] CPU port: {t_port:.1f} s for {n_rollouts} x ({N} x {T} env steps + train())")
    return rows


def test_configs1_forward_step_full_size():
    """BASELINE configs[1] exactly as bench.py runs it: HCWithPos, 64 envs x 2048 steps, batch 64, 10 epochs, target_kl 0.01,
    lr 3e-4 — 2 rollouts + 2 train() = up to 40 960 dependent optimiser steps."""
    agent, port, env = _pair("HCWithPos-v0", "hc", 64, 2048, 18, 6, [20], 0, batch_size=64, n_epochs=10, target_kl=0.01,
                             penalty_learning_rate=0.1)
    rows = _forward_step(agent, port, env, 2, 3e-4, 0.01, calibrated="configs1")
    assert rows[0]["steps"] > 2048            # at least one whole epoch ran at full size


def test_configs2_widths_long_chain():
    """BASELINE configs[2] flags (README.md:50: AntWall, constraint net [40, 40], batch 128 -> the two-workgroup update, 20 epochs,
    lr 3e-5, clip 0.4, lambdas 0.9, nu0 0.1, nu-lr 0.05, target_kl 0.02) at 256 envs x 128 steps: up to 20 x 256 = 5 120 dependent
    steps per train(), 2 rollouts (the target-KL test ends the loops after 3 072 + 1 280 steps).  Bounds calibrated port-vs-port on this
    very schedule (21 disturbances, profiles/r05_drift_calibration.md): see ANT_* above."""
    agent, port, env = _pair("AntWall-v0", "ant", 256, 128, 113, 8, [40, 40], 3, batch_size=128, n_epochs=20, target_kl=0.02,
                             learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, cost_gae_lambda=0.9,
                             penalty_initial_value=0.1, penalty_learning_rate=0.05)
    _forward_step(agent, port, env, 2, 3e-5, 0.02, calibrated="configs2")


# ---- round 6: the FULL row counts of BASELINE configs[2] and of the per-GPU shards of configs[3] / configs[4] -------------------------------
# (VERDICT r5 missing #3).  T = 2048 rows per env: 524 288 - 1 048 576 rows per rollout — PPOLagrangian._train_epochwise (one launch per epoch
# from 262 144 rows on, np.random.permutation drawn beside the running epoch), the 48 B / step schedule tables at up to 81 920 steps, storage
# offsets into 118 M-float observation planes and rollout_wide_kernel over 2048 steps; until round 6 these shapes had only run inside bench.py.
# One rollout + one train() each, against oracle.loop.PortAgent on the same noise and permutations (ref: ppo_lag.py:196-299,
# buffers.py:594-627, on_policy_algorithm.py:340-421).  Bounds CALIBRATED port-vs-port on THESE schedules (tools/calibrate_drift.py
# `configs2full`, `configs3shard`, `configs4shard`; profiles/r06_drift_calibration.md), 2 x the largest figure:
#   configs3shard (HC, lr 3e-4, all 81 920 steps run): the smooth chaotic regime of configs[1] — 12 disturbances (ulp moves, reversed / rotated minibatch
#     rows, a 1e-6-relative tanh error) give max |dp| = 6.1-9.0e-4 x lr x steps; HIP on MI355X: 9.8e-4.
#   configs2full / configs4shard (AntWall widths, lr 3e-5, clip 0.4; the target-KL test ends the loop after 2 / 1 epochs = 8 192 steps): the drift is a
#     COUNT of discrete events — one sample crossing the clip boundary moves max |dp| by a fixed amount (configs4shard: exactly 1.44e-4 x lr x steps in
#     5 of 8 ulp-sized disturbances, 1e-6 in the other 3; configs2full: exactly 4.1e-3 in 3 of 12, 1e-6 in the other 9), and how many samples cross depends
#     on the size of the disturbance: with a 1e-6-relative error in every tanh (the documented size of the kernels' v_exp-based tanh against libm's;
#     variants tanhe6<k>) 2.2-4.2e-3 (configs4shard) / 1.2-5.2e-3 (configs2full).  HIP on MI355X: 8.2e-4 / 3.4e-5 — between the two classes.
# The value losses of the two AntWall-width schedules get half the kernel-level tolerance instead (v_exp-based tanh against libm's: not a drift), as in
# test_configs2_widths_long_chain.  nu / average_cost: one rollout, collected with identical parameters — two float32 ulps of their values.
FULL_ROWS = {
    #                    max |dp| / (lr x steps)   pg loss   value losses (abs | None -> ANT_LOSS_RTOL)   average_cost   nu
    "configs2full": dict(dev=1.04e-2, pg=5.6e-6, vl=None, average_cost=1.2e-7, nu=1.5e-8),        # 2 x 5.18e-3 | 2 x 2.8e-6
    "configs3shard": dict(dev=1.8e-3, pg=2.4e-6, vl=1.7e-5, average_cost=1.2e-7, nu=2.4e-7),      # 2 x 8.85e-4 | 2 x 1.2e-6 | 2 x 8.4e-6
    "configs4shard": dict(dev=8.5e-3, pg=1.9e-6, vl=None, average_cost=1.2e-7, nu=2.4e-7),        # 2 x 4.24e-3 | 2 x 9.5e-7
}


def _full_rows_step(agent, port, env, name, lr, monkeypatch):
    """one rollout + one train() of both sides at the schedule `name`; the HIP side runs the way bench.py runs it — no stream object, the
    update draws np.random.permutation itself (here: patched to hand out SeededStreams(77)'s permutations, the ones the calibration used)."""
    from icrl_amd import logger
    from icrl_amd.ppo_lag import PPOLagrangian
    T, N, B = agent.n_steps, agent.n_envs, int(agent.batch_size)
    rows = T * N
    assert rows >= PPOLagrangian.LAZY_PERM_ROWS and agent.streams is None
    streams, pstreams = SeededStreams(77), SeededStreams(77)
    noise = streams.rollout_noise(T, N, port.act_dim)
    drawn, epochwise_calls = [], []
    monkeypatch.setattr(np.random, "permutation", lambda n: (drawn.append(len(drawn)), np.asarray(streams.permutation(drawn[-1], n)))[1])
    orig = agent._train_epochwise
    monkeypatch.setattr(agent, "_train_epochwise", lambda: (epochwise_calls.append(1), orig())[1])
    agent._setup_learn(rows)
    t0 = time.time()
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=torch.as_tensor(noise, device="cuda"))
    agent.train()
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    monkeypatch.undo()
    lg = dict(logger.Logger.CURRENT.name_to_value)
    assert epochwise_calls == [1], "the update did not take the epoch-wise path"
    executed = min(int(lg["train/early_stop_epoch"]) + 1, agent.n_epochs)
    assert executed <= len(drawn) <= min(executed + 1, agent.n_epochs), (executed, len(drawn))       # at most one permutation drawn in vain
    port.num_timesteps = 0
    port._last_obs = port.stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = port.stack.old_obs.copy()
    pstreams.rollout_noise(T, N, port.act_dim)
    t0 = time.time()
    b = port.collect_rollouts(noise)
    out = port.train(lambda e: pstreams.permutation(e, rows))
    t_port = time.time() - t0
    steps = agent.policy.adam_step
    assert steps == int(next(iter(port.optimizer.state.values()))["step"]) == executed * (-(-rows // B))
    # ---- the rollout: [T, N] planes of both sides (the float64 env / normaliser arithmetic is bit-exact; the MLP outputs are fp32)
    rb = agent.rollout_buffer
    buf_dev = {f: float(np.abs(getattr(rb, f).cpu().numpy().reshape(getattr(b, f).shape) - getattr(b, f)).max())
               for f in ("rewards", "costs", "orig_costs", "log_probs", "reward_values", "reward_advantages", "cost_advantages", "reward_returns", "cost_returns")}
    i_last = (T - 1, N - 1)
    assert np.allclose(rb.observations[T - 1].cpu().numpy(), b.observations[T - 1], rtol=0, atol=5e-5), "last row of the observation plane"
    assert np.array_equal(rb.dones.cpu().numpy().reshape(T, N), b.dones), "episode boundaries"
    for f, tol in (("rewards", 2e-5), ("orig_costs", 2e-5), ("log_probs", 2e-4), ("reward_values", 2e-4), ("reward_advantages", 5e-4), ("cost_advantages", 5e-4)):
        assert buf_dev[f] <= tol, (f, buf_dev[f])
    # ---- the update
    worst_abs = max(float(np.abs(v.numpy() - port.policy.params[k].detach().numpy()).max()) for k, v in agent.policy.state_dict().items())
    d = lambda key: abs(float(lg[key]) - float(out[key]))
    print(f"[{name}] {N} envs x {T} rows = {rows} rows, batch {B}: {steps} optimiser steps ({executed} of {agent.n_epochs} epochs, {len(drawn)} permutations drawn); "
          f"HIP {t_hip:.2f} s, CPU port {t_port:.1f} s; nu {lg['train/nu']:.9f} vs {out['train/nu']:.9f}; average_cost d {d('train/average_cost'):.2e}; "
          f"losses d pg {d('train/policy_gradient_loss'):.2e} rv {d('train/reward_value_loss'):.2e} cv {d('train/cost_value_loss'):.2e} "
          f"(values {out['train/policy_gradient_loss']:.3g} / {out['train/reward_value_loss']:.3g} / {out['train/cost_value_loss']:.3g}); approx_kl d {d('train/approx_kl'):.2e}; "
          f"explained variances d {d('train/reward_explained_variance'):.2e} / {d('train/cost_explained_variance'):.2e}; "
          f"max |d param| {worst_abs:.3e} = {worst_abs / (lr * steps):.2e} x lr x steps; buffer: " + ", ".join(f"{f} {v:.1e}" for f, v in buf_dev.items()))
    bd = FULL_ROWS[name]
    assert int(lg["train/early_stop_epoch"]) == int(out["train/early_stop_epoch"]), (lg["train/early_stop_epoch"], out["train/early_stop_epoch"], agent.epoch_kls, out.get("epoch_kls"))
    assert d("train/nu") <= bd["nu"] and d("train/average_cost") <= bd["average_cost"]
    assert worst_abs <= (bd["dev"] or ANT_DEV_BOUND) * lr * steps + 1e-9, (worst_abs, steps)
    for key in ("train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss"):
        tol = bd["pg" if "policy" in key else "vl"]
        assert d(key) <= (tol if tol is not None else 2e-8 + ANT_LOSS_RTOL * abs(out[key])), (key, lg[key], out[key])
    for key in ("train/reward_explained_variance", "train/cost_explained_variance"):
        assert d(key) <= 2e-5 + 1e-4 * abs(out[key]), (key, lg[key], out[key])
    return lg, out


def _pair_rows(env_id, kind, N, T, od, ad, cn_spec, seed, broken=False, **kw):
    """_pair without a stream object on the HIP side (so that train() takes the path bench.py takes)."""
    from icrl_amd import utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    env = utils.make_train_env(env_id, None, True, seed, N, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    if isinstance(cn_spec, str):          # the reference's AntBroken checkpoint through its off-by-one load() (constraint_net.py:394-399)
        cn = ConstraintNet.load(os.path.join(HERE, "golden", cn_spec))
        ocn = o_nets.CostNet(od, ad, [40, 40], False, None, None, None, None, None)
    else:
        lo = -np.ones(ad, np.float32)
        torch.manual_seed(seed + 1)
        cn = ConstraintNet(od, ad, cn_spec, None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
        ocn = o_nets.CostNet(od, ad, cn_spec, False, None, None, 20, lo, -lo)
    ocn.load_state_dict(cn.state_dict())
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=seed, **kw)
    stack = o_loop.make_stack(N, kind, seed, broken=broken); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, seed=seed, **kw)
    port.policy.load_state_dict(agent.policy.state_dict())
    return agent, port, env


def test_configs2_full_rows(monkeypatch):
    """BASELINE configs[2] at its full row count: AntWall, 256 envs x 2048 steps = 524 288 rows, README.md:50 flags (batch 128 -> two
    workgroups per network, 20 epochs of 4 096 minibatches, lr 3e-5, clip 0.4, lambdas 0.9, target_kl 0.02), constraint net [40, 40]."""
    agent, port, env = _pair_rows("AntWall-v0", "ant", 256, 2048, 113, 8, [40, 40], 3, batch_size=128, n_epochs=20, target_kl=0.02,
                                  learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, cost_gae_lambda=0.9,
                                  penalty_initial_value=0.1, penalty_learning_rate=0.05)
    _full_rows_step(agent, port, env, "configs2full", 3e-5, monkeypatch)


def test_configs3_shard_full_rows(monkeypatch):
    """BASELINE configs[3] as ONE GPU sees it: HCWithPos, 2048 / 8 = 256 envs x 2048 steps = 524 288 rows, README.md:38 flags (batch 64,
    10 epochs of 8 192 minibatches = up to 81 920 dependent optimiser steps, target_kl 0.01)."""
    agent, port, env = _pair_rows("HCWithPos-v0", "hc", 256, 2048, 18, 6, [20], 0, batch_size=64, n_epochs=10, target_kl=0.01,
                                  penalty_learning_rate=0.1)
    lg, out = _full_rows_step(agent, port, env, "configs3shard", 3e-4, monkeypatch)
    assert agent.policy.adam_step > 8192          # more than one whole epoch ran at full size


def test_configs4_shard_full_rows(monkeypatch):
    """BASELINE configs[4] as ONE GPU sees it: AntWallBroken, 4096 / 8 = 512 envs x 2048 steps = 1 048 576 rows, the reference's frozen
    AntBroken constraint net, README.md:78 flags (batch 128, 20 epochs of 8 192 minibatches, lr 3e-5, clip 0.4, reward lambda 0.9,
    target_kl 0.01, nu learning rate 1.0)."""
    agent, port, env = _pair_rows("AntWallBroken-v0", "ant", 512, 2048, 113, 8, "cn_antbroken.npz", 4, broken=True, batch_size=128, n_epochs=20,
                                  target_kl=0.01, learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, penalty_learning_rate=1.0)
    _full_rows_step(agent, port, env, "configs4shard", 3e-5, monkeypatch)


def test_configs4_shard_whole_learn_vs_port_band(golden):
    """BASELINE configs[4] as ONE GPU sees it, over a whole learn(): AntWallBroken-v0, 512 envs x 2048 steps, the reference's frozen AntBroken constraint
    net, README.md:78 flags, FOUR rollouts + updates = 4.2 M env steps (icrl/cpg.py:203 is one learn() call; the Lagrange multiplier moves with nu learning
    rate 1.0) against tests/golden/g22_whole_run_cpg.npz: what PortAgent's train() logged per rollout in 8 runs of the CPU port on the same SeededStreams(31)
    and initial weights — undisturbed and with ulp-sized / summation-order / 1e-6-tanh disturbances (tools/gen_whole_run.py run3 / band3).  Every scalar of
    every rollout must lie in [lo - w - tol, hi + w + tol], w = hi - lo (ref: ppo_lag.py:196-338, on_policy_algorithm.py:430-492)."""
    from icrl_amd import logger, utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    g = golden("g22_whole_run_cpg")
    N, T, seed = int(g["N"]), int(g["T"]), int(g["seed"])
    env = utils.make_train_env("AntWallBroken-v0", None, True, seed, N, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    cn = ConstraintNet.load(os.path.join(HERE, "golden", "cn_antbroken.npz"))
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=seed, streams=SeededStreams(int(g["stream_seed"])), batch_size=128, n_epochs=20, target_kl=0.01,
                          learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, penalty_learning_rate=1.0)
    agent.policy.load_state_dict({k[len("w0/"):]: g[k] for k in g.files if k.startswith("w0/")})
    keys = [str(k) for k in g["metric_keys"]]
    base, lo, hi = g["base"], g["lo"], g["hi"]
    n_it = base.shape[0]
    assert n_it >= 4 and {"train/nu", "train/average_cost", "train/early_stop_epoch", "train/policy_gradient_loss"} <= set(keys)
    agent._setup_learn(n_it * N * T)
    outside = []
    for it in range(n_it):
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        agent.train()
        lg = dict(logger.Logger.CURRENT.name_to_value)
        missing = [k for k in keys if k not in lg]
        assert not missing, missing
        for j, k in enumerate(keys):
            x, b, l, h = float(lg[k]), float(base[it, j]), float(lo[it, j]), float(hi[it, j])
            if k.endswith("explained_variance") and max(x, b, l, h) < 1.0:
                x, b, l, h = (1.0 / (1.0 - v) for v in (x, b, l, h))
            tol = 0.0 if k in ("train/early_stop_epoch", "train/n_updates") else (1e-6 if k == "train/nu" else 1e-5 + 1e-4 * abs(b))
            if k in ("train/approx_kl", "train/clip_fraction"):
                tol = 2e-4 + 2e-3 * abs(b)
            if not (l - (h - l) - tol <= x <= h + (h - l) + tol):
                outside.append((it, k, x, b, l, h))
        j = keys.index
        print(f"[cpg whole learn] rollout {it}: nu {lg['train/nu']:.6f} [{lo[it, j('train/nu')]:.6f}, {hi[it, j('train/nu')]:.6f}]  average_cost {lg['train/average_cost']:.6f} "
              f"[{lo[it, j('train/average_cost')]:.6f}, {hi[it, j('train/average_cost')]:.6f}]  early_stop_epoch {lg['train/early_stop_epoch']} "
              f"[{lo[it, j('train/early_stop_epoch')]:.0f}, {hi[it, j('train/early_stop_epoch')]:.0f}]")
    assert not outside, outside[:20]
