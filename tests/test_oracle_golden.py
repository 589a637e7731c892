"""CPU: the oracle restatement replayed against the golden vectors produced by the reference itself
(oracle/gen_golden.py).  Bit-exact where the arithmetic is numpy / identical torch ops."""
import os

import numpy as np
import pytest
import torch as th

from oracle import cn as o_cn, gae as o_gae, nets as o_nets, ppo as o_ppo, stats as o_stats
from oracle import loop as o_loop

HERE = os.path.dirname(os.path.abspath(__file__))


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


@pytest.mark.parametrize("case", ["t8n3", "t2000n1", "t256n16", "t1n4", "t64n5"])
def test_g1_gae_bit_exact(golden, case):
    g = _sub(golden("g1_gae"), case + "/")
    gr, lr_, gc, lc = g["params"]
    o = o_gae.dual_gae(g["rewards"], g["costs"], g["reward_values"], g["cost_values"], g["dones"],
                       g["last_v_r"], g["last_v_c"], g["last_dones"], gr, lr_, gc, lc)
    for k in o:
        assert o[k].dtype == np.float32
        assert np.array_equal(o[k], g[k]), k


def test_gae_f64_accumulation_matters():
    """The scan must carry float64: an all-f32 scan differs on long horizons."""
    rng = np.random.RandomState(0)
    T, N = 2000, 4
    r, v = rng.randn(T, N).astype(np.float32), rng.randn(T, N).astype(np.float32)
    d = np.zeros((T, N), np.float32)
    ret, adv = o_gae.gae_scan(r, v, d, v[-1], np.zeros(N, bool), 0.99, 0.95)
    acc, adv32 = np.zeros(N, np.float32), np.zeros((T, N), np.float32)
    for t in range(T - 1, -1, -1):
        nv = v[-1] if t == T - 1 else v[t + 1]
        acc = (r[t] + np.float32(0.99) * nv - v[t]) + np.float32(0.99 * 0.95) * acc
        adv32[t] = acc
    assert not np.array_equal(adv, adv32) and np.allclose(adv, adv32, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", ["hc", "ant", "lgw", "antbroken"])
def test_g2_cost_function(golden, case):
    g = _sub(golden("g2_cost_function"), case + "/")
    disc = case == "lgw"
    od, ad = g["obs"].shape[1], (2 if disc else g["acs"].shape[1])
    if case == "antbroken":       # ConstraintNet.load's positional shift: no clipping at all (ref: constraint_net.py:394-399)
        net = o_nets.CostNet(od, ad, list(g["hidden"]), False, None, None, None, None, None)
    else:
        lo = None if disc else -np.ones(ad, np.float32)
        net = o_nets.CostNet(od, ad, list(g["hidden"]), disc, None, None, 20, lo, None if disc else -lo)
    assert net.select_dim == list(g["select_dim"])
    net.load_state_dict({k[2:]: g[k] for k in g if k.startswith("w/")})
    assert np.array_equal(net.cost_function(g["obs"], g["acs"]), g["cost"])


def test_g2_select_dim_quirk():
    """acs_select_dim=None appends range(acs_dim), i.e. it re-selects the first obs columns, never the actions."""
    net = o_nets.CostNet(18, 6, [20])
    assert net.select_dim == list(range(18)) + list(range(6))


def test_g3_vecnormalize_stream(golden):
    g = golden("g3_vecnormalize")
    S, N, D = g["obs"].shape
    st = o_stats.NormState(N, D, reward_gamma=float(g["gammas"][0]), cost_gamma=float(g["gammas"][1]))
    assert np.array_equal(o_stats.norm_reset(st, g["reset_obs"]), g["reset_obs_n"])
    for t in range(S):
        o, r, c = o_stats.norm_step(st, g["obs"][t], g["rew"][t], g["costs"][t], g["done"][t])
        assert np.array_equal(o, g["obs_n"][t]) and np.array_equal(r, g["rew_n"][t]) and np.array_equal(c, g["cost_n"][t])
        assert np.array_equal(st.obs_rms.mean, g["obs_mean"][t]) and np.array_equal(st.obs_rms.var, g["obs_var"][t])
        assert st.ret_rms.var == g["ret_var"][t] and st.cost_rms.var == g["cost_var"][t]
        assert st.obs_rms.count == g["obs_count"][t] and st.cost_rms.count == g["cost_count"][t]


G13_WIDTHS = dict(policy_net=(32, 48), value_net=(64, 32), cost_value_net=(16, 64))      # oracle/gen_golden.py: g13_widths


G17 = dict(hidden=dict(policy_net=(64, 32, 32), value_net=(40,), cost_value_net=()), shared=(48,))      # gen_golden.py: g17_trunk
G18 = dict(hidden=dict(policy_net=(32,), value_net=(64, 64, 64), cost_value_net=(96, 200, 64, 16)))     # gen_golden.py: g18_deep


@pytest.mark.parametrize("name,hidden", [("g4_ppo_minibatch", (64, 64)), ("g13_widths", G13_WIDTHS), ("g17_trunk", G17), ("g18_deep", G18)])
def test_g4_ppo_minibatch(golden, name, hidden):
    """3 optimiser steps of the reference's own policy / optimizer objects; g13: widths other than 64, different per branch;
    g17: a shared trunk and branches of 3 / 1 / 0 layers; g18: branches of 1 / 3 / 4 layers (torch_layers.py:129-254)."""
    g = golden(name)
    pol = o_nets.TwoCriticPolicy(18, 6, **(hidden if isinstance(hidden, dict) and "hidden" in hidden else dict(hidden=hidden)))
    pol.load_state_dict(_sub(g, "w0/"))
    opt = th.optim.Adam(pol.parameters(), lr=float(g["lr"]), eps=1e-5)
    t = lambda k: th.as_tensor(g[k])
    for s in range(3):
        loss, tr = o_ppo.minibatch_loss(pol, t("obs"), t("act"), t("old_lp"), t("adv_r"), t("adv_c"), t("ret_r"),
                                        t("ret_c"), t("ret_r"), t("ret_c"), float(g["nu"]), float(g["clip"]))
        assert loss.item() == g[f"s{s}/loss"].item()
        assert tr["policy_loss"].item() == g[f"s{s}/policy_loss"].item()
        assert tr["approx_kl"].item() == g[f"s{s}/approx_kl"].item()
        assert tr["clip_fraction"].item() == g[f"s{s}/clip_fraction"].item()
        opt.zero_grad(); loss.backward()
        for k, p in pol.params.items():
            assert np.array_equal(p.grad.numpy(), g[f"s{s}/grad/{k}"]), k
        total, coef = o_ppo.clip_coef_explicit([p.grad for p in pol.parameters()], 0.5)
        assert abs(total - g[f"s{s}/grad_norm"].item()) <= 1e-5 * max(1.0, total)
        th.nn.utils.clip_grad_norm_(pol.parameters(), 0.5); opt.step()
        for k, p in pol.params.items():
            assert np.array_equal(p.detach().numpy(), g[f"s{s}/after/{k}"]), k


def test_explicit_adam_and_clip_match_torch():
    """The published Adam / clip_grad_norm_ formulas the HIP kernels implement == torch's (fp32 tolerance)."""
    th.manual_seed(0)
    p = th.randn(257, requires_grad=True)
    opt = th.optim.Adam([p], lr=3e-4, eps=1e-5)
    pe, m, v = p.detach().clone(), th.zeros(257), th.zeros(257)
    for step in range(1, 6):
        g = th.randn(257) * 3
        total, coef = o_ppo.clip_coef_explicit([g], 0.5)
        p.grad = g.clone()
        tn = th.nn.utils.clip_grad_norm_([p], 0.5)
        assert abs(float(tn) - total) < 1e-5
        opt.step()
        pe, m, v = o_ppo.adam_step_explicit(pe, g * coef, m, v, step, 3e-4, eps=1e-5)
        assert th.allclose(pe, p.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", "abcd")
def test_g5_dual_trajectory(golden, case):
    g = _sub(golden("g5_dual"), case + "/")
    nu0, lr, budget = (float(x) for x in g["params"])     # python floats, as the reference passes them
    d = o_ppo.Dual(budget, lr, nu0, None)
    for c, (nu, loss, log_nu) in zip(g["costs"], g["traj"]):
        d.update(c)
        assert d.nu().item() == nu and d.log_nu.item() == log_nu


def test_dual_clamp_floor_quirk():
    """nu0 = 1: log_nu floor is inv_softplus(inv_softplus(1)) -> nu floor ~ 0.5413."""
    d = o_ppo.Dual(10.0, 1.0, 1.0, None)          # huge budget: nu is driven down to the clamp
    for _ in range(50):
        d.update(0.0)
    assert abs(d.nu().item() - 0.5413) < 1e-3


@pytest.mark.parametrize("case", ["psis", "episode", "overflow", "earlystop"])
def test_g6_constraint_net_train(golden, case):
    g = _sub(golden("g6_constraint_net"), case + "/")
    psis, iters, tk_on, tk_no, lr = g["cfg"]
    lo = -np.ones(6, np.float32)
    net = o_nets.CostNet(18, 6, [20], False, None, None, 20, lo, -lo)
    net.load_state_dict(_sub(g, "w0/"))
    opt = th.optim.Adam(net.parameters(), lr=float(lr), eps=1e-5)
    m = o_cn.cn_train(net, opt, int(iters), net.prepare(g["nom_obs"], g["nom_acs"]), net.prepare(g["exp_obs"], g["exp_acs"]),
                      g["lengths"], reg_coeff=0.5, per_step=bool(psis), target_kl_old_new=float(tk_on),
                      target_kl_new_old=float(tk_no))
    for k, v in m.items():
        ref = g["m/" + k.split("/")[1]].item()
        assert (np.isnan(ref) and np.isnan(v)) or ref == v, k
    for k, p in net.params.items():
        assert np.array_equal(p.detach().numpy(), g["w1/" + k]), k


def test_g6_is_weight_shapes(golden):
    g = golden("g6_constraint_net")
    po, pn = th.as_tensor(g["isw/po"]), th.as_tensor(g["isw/pn"])
    for psis in (0, 1):
        w, a, b = o_cn.is_weights_and_kls(po, pn, np.array([10, 20]), 1e-5, bool(psis))
        assert tuple(w.shape) == ((30, 1) if psis else (30,))
        assert np.array_equal(w.numpy(), g[f"isw{psis}/w"]) and [a.item(), b.item()] == list(g[f"isw{psis}/kl"])


def test_psis_broadcast_quirk_equals_product_of_means():
    th.manual_seed(1)
    w, logp = th.rand(40, 1) + 0.5, th.randn(40, 1)
    full = th.mean(w[np.arange(40)][..., None] * logp)
    assert abs(full.item() - (w.mean() * logp.mean()).item()) < 1e-6


def test_g9_learn_iteration_teacher_forced(golden):
    """A full learn() (2 rollouts + 2 train()) of the CPU port == the reference's, given the recorded streams."""
    g = golden("g9_learn_iteration")
    noise = g["noise"]
    _, T, N, A = noise.shape
    stack = o_loop.make_stack(N, "hc", 0)
    lo = -np.ones(6, np.float32)
    cn = o_nets.CostNet(18, 6, [20], False, None, None, 20, lo, -lo)
    cn.load_state_dict(_sub(g, "cn/"))
    stack.cost_fn = cn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, seed=0)
    port.policy.load_state_dict(_sub(g, "w0/"))
    perms = [g["perms0"], g["perms1"]]
    port.learn(2 * N * T, noise_fn=lambda it: noise[it], perms_fn=lambda it: (lambda e, p=perms[it]: p[e]))
    for k in ("observations", "orig_observations", "actions", "rewards", "costs", "orig_costs", "dones", "log_probs",
              "reward_values", "cost_values", "reward_advantages", "cost_advantages", "reward_returns", "cost_returns"):
        assert np.array_equal(getattr(port.buf, k).reshape(T, N, -1), g["buf/" + k]), k
    for k, p in port.policy.params.items():
        assert np.array_equal(p.detach().numpy(), g["w1/" + k]), k
    assert port.logs["train/nu"] == g["log/nu"].item()
    assert np.array_equal(stack.norm.obs_rms.mean, g["obs_rms_mean"]) and stack.norm.cost_rms.var == g["cost_rms_var"]
    # nominal sampling: (s_{t+1}, a_t) pairing, clipped actions, auto-reset observation at episode ends
    s1 = o_loop.make_stack(1, "hc", 0, training=False, norm_reward=False, norm_cost=False)
    o_loop.sync_normalization(stack.norm, s1.norm)
    port.stack = s1
    oo, o, a, r, l = o_loop.sample_from_agent(port, s1, 2, g["sample_noise"])
    assert np.array_equal(l, g["sample_lengths"]) and np.array_equal(oo, g["sample_orig_obs"])
    assert np.array_equal(a, g["sample_actions"]) and np.allclose(r, g["sample_rewards"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("env_id", ["lgw", "clgw"])
def test_g10_lap_grid_env(golden, env_id):
    """LapGridWorld / ConstrainedLapGridWorld restatement == the reference's gym envs behind DummyVecEnv (auto-reset)."""
    from oracle.lap_grid import LapGridVecEnv
    g = golden("g10_lap_grid")
    acts = g[env_id + "/actions"]
    env = LapGridVecEnv(acts.shape[1], constrained=(env_id == "clgw"))
    assert np.array_equal(env.reset(), -np.ones((acts.shape[1], 1)))
    for t in range(acts.shape[0]):
        o, r, d = env.step(acts[t])
        assert np.array_equal(o, g[env_id + "/obs"][t]) and np.array_equal(r, g[env_id + "/rew"][t]) and np.array_equal(d, g[env_id + "/done"][t])


def test_g10_discrete_learn_teacher_forced(golden):
    """configs[0] shapes: Categorical policy, one-hot cost net, normalisation off — port == reference bit for bit."""
    g = golden("g10_lap_grid")
    acts = g["learn_actions"]
    _, T, N = acts.shape
    stack = o_loop.make_stack(N, "lgw", 0, norm_obs=False, norm_reward=False, norm_cost=False)
    cn = o_nets.CostNet(1, 2, [20], True, None, None, 20, None, None)
    cn.load_state_dict(_sub(g, "cn/"))
    stack.cost_fn = cn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, seed=3, discrete=True, ent_coef=0.01)
    port.policy.load_state_dict(_sub(g, "w0/"))
    perms = [g["perms0"][:g["n_perms"][0]], g["perms1"][:g["n_perms"][1]]]
    port.learn(2 * N * T, noise_fn=lambda it: acts[it].astype(np.float32), perms_fn=lambda it: (lambda e, p=perms[it]: p[e]))
    for k in ("observations", "actions", "rewards", "costs", "orig_costs", "dones", "log_probs", "reward_values", "cost_values",
              "reward_advantages", "cost_advantages", "reward_returns", "cost_returns"):
        assert np.array_equal(getattr(port.buf, k).reshape(T, N, -1), g["buf/" + k]), k
    for k, p in port.policy.params.items():
        assert np.array_equal(p.detach().numpy(), g["w1/" + k]), k
    assert port.logs["train/nu"] == g["log/nu"].item()
    assert port.logs["train/entropy_loss"] == pytest.approx(g["log/entropy_loss"].item(), abs=1e-7)


@pytest.mark.parametrize("case", ["mb_psis", "mb_episode", "mb_nois", "mb_gail"])
def test_g7_constraint_net_minibatch(golden, case):
    g = _sub(golden("g7_constraint_net_minibatch"), case + "/")
    psis, iters, tk_on, tk_no, lr, bs, nois, gail = g["cfg"]
    lo = -np.ones(6, np.float32)
    net = o_nets.CostNet(18, 6, [20], False, None, None, 20, lo, -lo)
    net.load_state_dict(_sub(g, "w0/"))
    opt = th.optim.Adam(net.parameters(), lr=float(lr), eps=1e-5)
    replay = iter(g["perms"])
    m = o_cn.cn_train(net, opt, int(iters), net.prepare(g["nom_obs"], g["nom_acs"]), net.prepare(g["exp_obs"], g["exp_acs"]),
                      g["lengths"], reg_coeff=0.5, per_step=bool(psis), target_kl_old_new=float(tk_on),
                      target_kl_new_old=float(tk_no), batch_size=int(bs), importance_sampling=not bool(nois), gail=bool(gail),
                      rng=type("R", (), {"permutation": staticmethod(lambda n: next(replay))}))
    for k, v in m.items():
        ref = g["m/" + k.split("/")[1]].item()
        assert (np.isnan(ref) and np.isnan(v)) or ref == v, k
    for k, p in net.params.items():
        assert np.array_equal(p.detach().numpy(), g["w1/" + k]), k


# what of the reference's per-iteration scalars is NOT compared, and why: wall clock; Monitor's episode statistics (the reference wraps every
# env in a Monitor, vec_env/subproc_vec_env + common/monitor.py — the device-resident env stack keeps no episode info buffer)
G8_SKIP = ("time(m)", "time/fps", "time/time_elapsed", "rollout/ep_len_mean", "rollout/ep_rew_mean")


def test_g8_icrl_outer_loop_reference(golden):
    """The reference's OWN icrl(config) on LGW-v0 / CLGW-v0 (3 outer iterations, real SubprocVecEnv workers; g8): the CPU port,
    teacher-forced with the recorded action / permutation draws, reproduces every per-iteration metric the reference logged
    (train/nu, train/average_cost, true/cost, true/reward, the two "KL"s, all backward/*) and the final weights."""
    from oracle.streams import RecordedStreams
    from icrl_amd.icrl import build_parser            # host-side flag parser only (no GPU use)
    g = golden("g8_icrl_lgw")
    ex = golden("expert_lgw")
    cfg = vars(build_parser().parse_args([str(a) for a in g["argv"]]))
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"][:4000], ex["actions"][:4000], _sub(g, "expert_policy/"),
                                          streams=RecordedStreams(g), init=dict(policy=_sub(g, "w0/"), cn=_sub(g, "cn0/")))
    keys = [str(k) for k in g["metric_keys"]]      # EVERY scalar the reference logged (icrl/icrl.py:283-300)
    assert {"forward/nu", "forward/average_cost", "true/cost", "true/reward", "true/forward_kl", "true/reverse_kl", "backward/cn_loss",
            "backward/kl_new_old", "backward/early_stop_itr", "forward/reward_explained_variance", "forward/cost_explained_variance",
            "time/fps", "time/time_elapsed", "time/iterations", "time/total_timesteps"} <= set(keys)
    missing = [k for k in keys if k not in om[0] and k not in G8_SKIP]
    assert not missing, missing
    assert steps == 3 * 800
    for it in range(3):
        for j, k in enumerate(keys):
            if k in G8_SKIP:
                continue
            ref, got = float(g["metrics"][it, j]), float(om[it][k])
            # identical torch / numpy ops in the same order: exact, up to the float64-vs-float32 mean of two logged averages
            assert got == ref or abs(got - ref) <= 2e-7 * max(1.0, abs(ref)), (it, k, got, ref)
    for k, v in _sub(g, "w1/").items():
        assert np.array_equal(objs["agent"].policy.params[k].detach().numpy(), v), k
    for k, v in _sub(g, "cn1/").items():
        assert np.array_equal(objs["cn"].params[k].detach().numpy(), v), k


@pytest.mark.parametrize("case", ["default_cpg", "derivative", "integral"])
def test_g11_pid_lagrangian(golden, case):
    """PIDLagrangian (cpg --use_pid): the oracle restatement AND the product's host-side controller against the reference's
    trajectories, bit for bit (pure float arithmetic in the same order)."""
    from icrl_amd.dual_variable import PIDLagrangian
    g = _sub(golden("g11_pid"), case + "/")
    names = ("alpha", "penalty_init", "Kp", "Ki", "Kd", "pid_delay", "delta_p_ema_alpha", "delta_d_ema_alpha")
    kw = {k: float(v) for k, v in zip(names, g["params"])}; kw["pid_delay"] = int(kw["pid_delay"])
    orc, prod = o_ppo.PID(**kw), PIDLagrangian(**kw)
    for c, (nu, loss, pid_i, dp, cd) in zip(g["costs"], g["traj"]):
        orc.update(c); prod.update_parameter(c)
        assert orc.nu().item() == nu and orc.loss.item() == loss
        assert prod.nu().item() == nu and prod.loss.item() == loss
        assert prod.pid_i == pid_i and prod._delta_p == dp and prod._cost_delta == cd


def test_reference_written_files_host_readers(golden, tmp_path):
    """files exactly as the REFERENCE wrote them (tests/golden/ref_artifacts): the expert-rollout pickle (icrl/icrl.py:25-43), the
    `data` JSON of an agent archive (save_util.py:72-176) and the pickled VecNormalizeWithCost statistics
    (vec_normalize.py:42-64) are read by the build's host code without stable_baselines3 / gym."""
    import os, shutil, zipfile
    from icrl_amd import utils
    art = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden/ref_artifacts")
    exp = golden("ref_artifacts_expected")
    d = tmp_path / "HC" / "files" / "EXPERT" / "rollouts"
    d.mkdir(parents=True)
    for i in range(2):
        shutil.copyfile(os.path.join(art, "hc_expert_rollout_0.pkl"), d / f"{i}.pkl")
    (obs, acs), mean_reward = utils.load_expert_data(str(tmp_path / "HC"), 2)
    hc = golden("expert_hc")
    assert obs.shape == (1000, 18) and obs.dtype == np.float64 and acs.shape == (1000, 6) and acs.dtype == np.float32
    assert np.array_equal(obs[:500].astype(np.float32), hc["observations"][:500]) and np.array_equal(acs[:500], hc["actions"][:500])
    assert np.isfinite(mean_reward)
    with zipfile.ZipFile(os.path.join(art, "hc_best_model.zip")) as z:
        data = utils.parse_sb3_data(z.read("data"))
    assert data["observation_space"].shape == (18,) and data["action_space"].shape == (6,)
    assert np.all(data["action_space"].low == -1) and np.all(np.isinf(data["observation_space"].high))
    assert data["num_timesteps"] == int(exp["num_timesteps"]) and data["_n_updates"] == int(exp["n_updates"]) and data["n_envs"] == int(exp["n_envs"])
    assert data["target_kl"] == 0.01 and data["algo_type"] == "lagrangian" and "lr_schedule" not in data and "dual" not in data
    vn = utils.load_reference_pickle(os.path.join(art, "hc_train_env_stats.pkl"))
    assert type(vn).__name__ == str(exp["vn_class"])
    assert np.array_equal(vn.obs_rms.mean, exp["obs_rms_mean"]) and np.array_equal(vn.obs_rms.var, exp["obs_rms_var"])
    assert vn.obs_rms.count == float(exp["obs_rms_count"]) and vn.ret_rms.var == float(exp["ret_rms_var"]) and vn.clip_obs == float(exp["clip_obs"])


def test_g12_gail_baseline_reference(golden):
    """GAIL-constraint baseline (icrl/gail_utils.py, icrl/gail.py): the CPU port — discriminator, rollout-end hook and plain PPO
    expressed through the two-critics agent with the cost path off — teacher-forced with the draws recorded from the reference's
    GailDiscriminator + GailCallback + PPO (g12).  The port differs from the reference by at most 1 ulp per quantity (the idle
    cost critic's zero gradients change the association of clip_grad_norm_'s sum), hence 1e-6, not bit equality."""
    from oracle import gail as o_gail
    g = golden("g12_gail")
    noise, perms, is_disc = g["noise"], g["perms"], g["perm_is_disc"]
    _, T, N, _ = noise.shape
    stack = o_loop.make_stack(N, "hc", 0, norm_cost=False)
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.02, seed=0, cost_vf_coef=0.0, penalty_initial_value=0.0)
    psd = {k: v.clone() for k, v in port.policy.state_dict().items()}
    psd.update({k: th.as_tensor(v) for k, v in _sub(g, "w0/").items()})
    port.policy.load_state_dict(psd)
    net = o_gail.make_disc(18, 6, [20])
    net.load_state_dict({k: th.as_tensor(v) for k, v in _sub(g, "d0/").items()})
    opt = th.optim.Adam(net.parameters(), lr=0.01, eps=1e-5)
    wall = lambda o, a: (o[..., 0] <= -0.05)
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    cursor = 0
    for it in range(2):
        port.collect_rollouts(noise[it])
        assert is_disc[cursor]
        replay = iter([perms[cursor]]); cursor += 1
        m = o_gail.rollout_end(port, net, opt, g["exp_obs"], g["exp_acs"], port.last_values, port.last_dones_out, batch_size=48,
                               true_cost_fn=wall, rng=type("R", (), {"permutation": staticmethod(lambda n: next(replay))}))
        for k in ("disc_loss", "expert_loss", "nominal_loss", "mean_nominal_preds", "mean_expert_preds"):
            assert abs(m["discriminator/" + k] - float(g[f"disc_metrics/{it}/{k}"])) < 1e-6, (it, k)
        assert m["eval/mean_cost"] == float(g["mean_cost"][it])
        assert np.allclose(port.buf.rewards, g["rewards"][it], rtol=0, atol=1e-6)
        assert np.allclose(port.buf.reward_advantages, g["advantages"][it], rtol=0, atol=5e-6)
        assert np.allclose(port.buf.reward_returns, g["returns"][it], rtol=0, atol=5e-6)
        remaining = perms[cursor:]
        res = port.train(lambda e, r=remaining: r[e])
        cursor += min(int(res["train/early_stop_epoch"]) + 1, 3)
    assert cursor == len(perms)
    for k, v in _sub(g, "w1/").items():
        assert np.allclose(port.policy.params[k].detach().numpy(), v, rtol=0, atol=1e-6), k
    for k, v in _sub(g, "d1/").items():
        assert np.allclose(net.params[k].detach().numpy(), v, rtol=0, atol=1e-6), k
    assert np.allclose(o_gail.disc_reward(net, g["probe_obs"], g["probe_acs"]), g["probe_reward"], rtol=0, atol=1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference exists in the build container only")
def test_reference_loads_the_archive_this_build_wrote():
    """tests/golden/hip_agent_archive.zip was written by `icrl_amd.PPOLagrangian.save` on an MI355X
    (tests/test_ref_artifacts_gpu.py::test_agent_archive_for_the_reference_loader); the REFERENCE's own `PPOLagrangian.load`
    (base_class.py:564-645) — imported unmodified under oracle/ref_shim — rebuilds the agent from it and its deterministic `predict`
    on 64 observations equals what the HIP path computed (hip_agent_archive_expected.npz)."""
    import subprocess, sys
    root = os.path.dirname(HERE)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle", "ref_shim"), "/root/reference", "/root/reference/custom_envs", root]))
    out = subprocess.run([sys.executable, "-W", "ignore", "-m", "oracle.verify_agent_archive", os.path.join(HERE, "golden", "hip_agent_archive.zip"),
                          os.path.join(HERE, "golden", "hip_agent_archive_expected.npz")], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "VERIFIED" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
