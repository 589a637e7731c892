"""GPU: trajectory-level parity of the ICRL outer loop (icrl/icrl.py:199-304): >= 3 outer iterations of
forward -> sample -> backward -> new cost function -> forward on the HIP path against
  (1) the metrics the REFERENCE's own icrl() logged on LGW-v0 / CLGW-v0 (tests/golden/g8_icrl_lgw.npz), and
  (2) the CPU port (oracle.loop.icrl_port, pinned by g8) on HCWithPos shapes,
both teacher-forced with the same action-noise / permutation streams (oracle/streams.py).

Tolerances (fp32 MLP arithmetic on the GPU vs torch-CPU; float64 env / statistics are bit-exact):
  train/nu                      1e-5 absolute (north_star's "Lagrange multiplier trajectory")
  train/average_cost, losses    1e-5 + 1e-4 relative
  true/cost                     exact on LGW (discrete actions are teacher-forced), one sample of the batch on HC
  true/reward                   exact on LGW, 1e-4 relative on HC
  backward/*                    the g6 tolerances of tests/test_cn_train_gpu.py (2e-3 relative + 2e-4)
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import loop as o_loop
from oracle.streams import RecordedStreams, SeededStreams

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


def _close(a, b, rtol, atol):
    if np.isnan(b) or np.isinf(b):
        return (np.isnan(a) and np.isnan(b)) or a == b
    return abs(a - b) <= atol + rtol * abs(b)


EXACT_INT = ("forward/early_stop_epoch", "backward/early_stop_itr", "forward/n_updates", "timesteps", "iteration", "time/iterations", "time/total_timesteps")
WALL_CLOCK = ("time(m)", "time/fps", "time/time_elapsed", "time/forward_s", "time/rest_s")      # never compared


def _compare(it, got, ref, keys, discrete, n_nominal):
    worst, bad = {}, []
    for k in keys:
        if k in WALL_CLOCK:
            continue
        a, b = float(got[k]), float(ref[k])
        if k in EXACT_INT:
            ok = a == b
        elif k == "forward/nu":
            ok = abs(a - b) <= 1e-5
        elif k in ("true/cost", "best_true/best_cost", "true/samples_behind", "true/samples_infront"):
            ok = a == b if discrete else abs(a - b) <= 1.0 / n_nominal + 1e-12
        elif k in ("true/reward", "true/reward_std", "best_true/best_reward"):
            ok = _close(a, b, 0.0 if discrete else 1e-4, 1e-9 if discrete else 1e-4)
        elif k.startswith("backward/"):
            ok = _close(a, b, 2e-3, 2e-4)
        elif k in ("forward/approx_kl", "forward/clip_fraction"):
            ok = _close(a, b, 2e-3, 2e-4)        # means of per-minibatch quantities of size ~1e-3 .. 1e-1
        elif k.endswith("explained_variance"):
            ok = _close(a, b, 1e-4, 2e-5)        # 1 - a ratio of two float32 variances (numpy: float32 pairwise sums; the kernel: float64 sums)
        else:
            ok = _close(a, b, 1e-4, 1e-5)
        if not ok:
            bad.append((it, k, a, b))
        if np.isfinite(b):
            worst[k] = max(worst.get(k, 0.0), abs(a - b))
    assert not bad, bad
    return worst


def test_icrl_lgw_three_iterations_vs_reference(golden):
    """HIP icrl() vs the reference's own icrl() run (g8): README.md:25 flags at the recorded reduced size."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    g = golden("g8_icrl_lgw")
    expert = os.path.join(HERE, "golden/expert_lgw.npz")
    argv = [str(a) for a in g["argv"]] + ["-ep", expert, "--expert_agent_path", expert, "-v", "0"]
    argv[argv.index("-d") + 1] = "cuda"
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1, streams=RecordedStreams(g))
    st = setup(types.SimpleNamespace(**cfg))
    # the reference's expert agent (expert_lgw.npz carries best_model.zip's policy.pth) must be what g8 recorded
    for k, v in _sub(g, "expert_policy/").items():
        assert np.array_equal(st["expert_agent"].policy.state_dict()[k].numpy(), v), k
    st["agent"].policy.load_state_dict(_sub(g, "w0/"))
    st["constraint_net"].load_state_dict(_sub(g, "cn0/"))
    # every scalar the reference logged, minus the explicit skip list: wall clock, and Monitor's episode statistics (the device-resident env
    # stack keeps no episode info buffer; SURVEY section 2 scopes Monitor out)
    skip = ("time(m)", "time/fps", "time/time_elapsed", "rollout/ep_len_mean", "rollout/ep_rew_mean")
    keys = [str(k) for k in g["metric_keys"] if str(k) not in skip]
    assert len(keys) >= 50 and {"forward/reward_explained_variance", "forward/cost_explained_variance", "time/iterations", "forward/learning_rate"} <= set(keys)
    worst = {}
    for it in range(3):
        m = outer_iteration(st, it)
        missing = [k for k in keys if k not in m]
        assert not missing, missing
        assert all(k in m for k in ("time/fps", "time/time_elapsed"))      # logged (wall clock: not compared)
        ref = dict(zip([str(k) for k in g["metric_keys"]], g["metrics"][it]))
        for k, v in _compare(it, m, ref, keys, True, 4000).items():
            worst[k] = max(worst.get(k, 0.0), v)
    print("worst absolute deviation from the reference over 3 outer iterations:",
          {k: float(f"{v:.3g}") for k, v in sorted(worst.items()) if v > 0})
    # final weights after 3 x (2 rollouts + 2 x <=10 epochs x 7 minibatches) Adam steps and 3 x 20 constraint-net iterations
    for k, v in st["agent"].policy.state_dict().items():
        assert np.allclose(v.numpy(), g["w1/" + k], rtol=1e-3, atol=2e-5), (k, np.abs(v.numpy() - g["w1/" + k]).max())
    for k, v in st["constraint_net"].state_dict().items():
        assert np.allclose(v.numpy(), g["cn1/" + k], rtol=2e-3, atol=2e-4), (k, np.abs(v.numpy() - g["cn1/" + k]).max())


def _lgw_full_size(golden, ft, n_iters):
    """HIP icrl() and the CPU port on BASELINE configs[0]'s shapes — LGW-v0 / CLGW-v0, 1 env, n_steps 2000 (minibatches 31 x 64 + 16), the
    reference's README.md:25 flags, `ft` forward timesteps per outer iteration — on the same uniform / permutation streams."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    expert = os.path.join(HERE, "golden/expert_lgw.npz")
    argv = ["icrl", "-er", "20", "-ep", expert, "--expert_agent_path", expert, "-tei", "LGW-v0", "-eei", "CLGW-v0", "-tk", "0.01", "-cl", "20",
            "-clr", "0.003", "-ft", str(ft), "-ni", str(n_iters), "-bi", "20", "-dno", "-dnr", "-dnc", "--n_steps", "2000", "-nt", "1", "-s", "0", "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1, streams=SeededStreams(5))
    st = setup(types.SimpleNamespace(**cfg))
    init = dict(policy={k: v.numpy().copy() for k, v in st["agent"].policy.state_dict().items()},
                cn={k: v.numpy().copy() for k, v in st["constraint_net"].state_dict().items()})
    ex = golden("expert_lgw")
    n_thr = torch.get_num_threads()
    torch.set_num_threads(1)          # the port's 64-wide MLP steps are fastest on one thread
    try:
        port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
        om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"], ex["actions"], _sub(ex, "policy/"), streams=SeededStreams(5), init=init)
    finally:
        torch.set_num_threads(n_thr)
    return st, om, steps, [outer_iteration(st, it) for it in range(n_iters)]


def test_icrl_lgw_full_size_vs_port(golden):
    """BASELINE configs[0] at FULL size (VERDICT r4 #4c): 1 env, n_steps 2000, README.md:25 flags, against the CPU port (pinned to the
    reference's own icrl() on this env by g8 at 2 envs x 200).  Discrete actions come from the inverse CDF of teacher-forced uniforms:
    as long as no uniform lands between the two sides' CDFs the env traces are IDENTICAL and everything is compared at the module's
    tolerances; the first flipped sample moves the rest of its episode, the updates fed with it and — LapGridWorld's policy swings
    hard (approx_kl up to 0.5 per update) — every later metric by per cents.  Measured on MI355X in round 5 (growing forward_timesteps):
    HIP and port agree exactly in all traces for 30 000 env steps = 15 rollouts + 15 updates (max |d param| 7.7e-6 after 4 800 Adam
    steps), and part somewhere in rollouts 16-25.  The port against ITSELF (5 runs with every initial parameter moved by -1 / 0 / +1
    ulp, tools/calibrate_lgw.py, profiles/r05_lgw_calibration.md) holds all of iteration 0 (forward metrics within 6e-5) and is
    chaotic in iteration 1 (early_stop_epoch 0 vs 10, true/cost 0.39 vs 0.20, nu +- 3.9e-3, average_cost +- 2.8e-2).  So:
      (A) here: one outer iteration with forward_timesteps 30 000: EVERY metric strict — forward, sampled episodes, constraint-net update, KLs;
      (B) the README's 0.5e5 over 5 outer iterations: test_icrl_whole_run_vs_port_band[g20_whole_run_lgw] (round 6: a band of 8 port runs instead of one
          pair of runs with hand-set bounds — VERDICT r5 weak #1b)."""
    st, om, steps, ms = _lgw_full_size(golden, 30000, 1)
    keys = sorted(k for k in om[0] if k in ms[0] and k not in ("forward/std",) and not k.startswith("time/"))
    worst = _compare(0, ms[0], om[0], keys, True, 4000)
    assert st["timesteps"] == steps == 15 * 2000 and ms[0]["forward/n_updates"] == om[0]["forward/n_updates"] == 150
    print("LGW full size (A), 15 rollouts + updates + sampling + constraint-net update: worst absolute deviation from the CPU port",
          {k: float(f"{v:.3g}") for k, v in sorted(worst.items()) if v > 0})


def test_icrl_hc_three_iterations_vs_port(golden):
    """HIP icrl() vs the CPU port on HCWithPos shapes (N = 8, T = 128), README.md:38 flags, 3 outer iterations, same
    noise / permutation streams.  The port itself is pinned to the reference by g8 (outer loop) and g9 (HC learn())."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "10", "-ft", "2000",
            "-ni", "3", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis",
            "-ctkno", "2.5", "-nt", "8", "--n_steps", "128", "-s", "3", "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1, streams=SeededStreams(11))
    st = setup(types.SimpleNamespace(**cfg))
    init = dict(policy={k: v.numpy().copy() for k, v in st["agent"].policy.state_dict().items()},
                cn={k: v.numpy().copy() for k, v in st["constraint_net"].state_dict().items()})
    ex = golden("expert_hc")
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"][:1000], ex["actions"][:1000], _sub(ex, "policy/"),
                                          streams=SeededStreams(11), init=init)
    keys = sorted(k for k in om[0] if k not in ("forward/std",))
    worst = {}
    for it in range(3):
        m = outer_iteration(st, it)
        for k, v in _compare(it, m, om[it], [k for k in keys if k in m], False, 2000).items():
            worst[k] = max(worst.get(k, 0.0), v)
        assert abs(m["forward/std"] - om[it]["forward/std"]) < 1e-5
    assert st["timesteps"] == steps == 3 * 2048
    print("worst absolute deviation from the CPU port over 3 outer iterations:",
          {k: float(f"{v:.3g}") for k, v in sorted(worst.items()) if v > 0})
    nus = [m_["forward/nu"] for m_ in om]
    assert len(set(np.round(nus, 4))) == 3          # the multiplier actually moves between iterations


def test_icrl_hc_wide_policy_two_iterations_vs_port(golden):
    """The same loop with `-pl 128 128 -rvl 96 128 -cvl 128 80` (icrl/utils.py:636-655): layers above 64 run on the generic-shape path —
    per-step rollouts and episode loops over the fine-grained entry points, policy_generic_kernel, four launches per optimiser step
    (csrc/generic.hip) — through the unchanged outer loop, against the CPU port with the same widths and streams."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "10", "-ft", "2000",
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis",
            "-ctkno", "2.5", "-nt", "8", "--n_steps", "128", "-s", "3", "-v", "0", "-pl", "128", "128", "-rvl", "96", "128", "-cvl", "128", "80"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1, streams=SeededStreams(12))
    st = setup(types.SimpleNamespace(**cfg))
    assert st["agent"].policy.wide and st["agent"].policy.hw == 128
    init = dict(policy={k: v.numpy().copy() for k, v in st["agent"].policy.state_dict().items()},
                cn={k: v.numpy().copy() for k, v in st["constraint_net"].state_dict().items()})
    ex = golden("expert_hc")
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"][:1000], ex["actions"][:1000], _sub(ex, "policy/"),
                                          streams=SeededStreams(12), init=init)
    assert tuple(objs["agent"].policy.params["mlp_extractor.value_net.0.weight"].shape) == (96, 18)
    keys = sorted(k for k in om[0] if k not in ("forward/std",))
    for it in range(2):
        m = outer_iteration(st, it)
        _compare(it, m, om[it], [k for k in keys if k in m], False, 2000)
        assert abs(m["forward/std"] - om[it]["forward/std"]) < 1e-5
    assert st["timesteps"] == steps == 2 * 2048


def test_icrl_hc_shared_trunk_two_iterations_vs_port(golden):
    """The same loop with `-sl 48 -pl 64 32 32 -rvl 40 -cvl` (icrl/utils.py:636-655, torch_layers.py:129-254): a shared trunk, branches of
    3 / 1 / 0 layers — the architectures icrl_policy_t.arch describes — through the unchanged outer loop (per-step rollouts and episode
    loops, the table-driven generic kernels) against the CPU port with the same architecture and streams; then save() / load().
    Stream seed 14: with this narrow, deep actor the loop is touchy — the port run against ITSELF from weights perturbed by 3e-5 jumps to
    1e-3 parameter distance within four updates for stream seeds 12, 13 and 15 (one sample crossing the clip boundary) and stays at 5e-5 for
    14; the product's own drift against the port reaches 2e-5 in the observations by then, so the comparison needs a seed in the smooth
    regime to say anything about the kernels."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "10", "-ft", "2000",
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis",
            "-ctkno", "2.5", "-nt", "8", "--n_steps", "128", "-s", "3", "-v", "0", "-sl", "48", "-pl", "64", "32", "32", "-rvl", "40", "-cvl"]
    cfg = vars(build_parser().parse_args(argv))
    assert cfg["shared_layers"] == [48] and cfg["cost_vf_layers"] == []
    cfg.update(rank=0, world_size=1, streams=SeededStreams(14))
    st = setup(types.SimpleNamespace(**cfg))
    pol = st["agent"].policy
    assert pol.wide and pol.kind == "arch" and pol.shared == (48,) and pol.layers["cost_value_net"] == ()
    init = dict(policy={k: v.numpy().copy() for k, v in pol.state_dict().items()},
                cn={k: v.numpy().copy() for k, v in st["constraint_net"].state_dict().items()})
    ex = golden("expert_hc")
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"][:1000], ex["actions"][:1000], _sub(ex, "policy/"),
                                          streams=SeededStreams(14), init=init)
    assert tuple(objs["agent"].policy.params["cost_value_net.weight"].shape) == (1, 48)      # the cost-value head reads the trunk
    assert list(objs["agent"].policy.params) == list(pol.shapes)
    keys = sorted(k for k in om[0] if k not in ("forward/std",))
    for it in range(2):
        m = outer_iteration(st, it)
        _compare(it, m, om[it], [k for k in keys if k in m], False, 2000)
        assert abs(m["forward/std"] - om[it]["forward/std"]) < 1e-5
    assert st["timesteps"] == steps == 2 * 2048
    # archive round trip: the architecture is read back off the stored tensors (base_class.py:564-645)
    import tempfile
    from icrl_amd.ppo_lag import PPOLagrangian
    with tempfile.TemporaryDirectory() as d:
        path = st["agent"].save(os.path.join(d, "model"))
        again = PPOLagrangian.load(path)
        assert again.policy.kind == "arch" and again.policy.shared == (48,) and again.policy.layers == pol.layers
        for k, v in pol.state_dict().items():
            assert torch.equal(v, again.policy.state_dict()[k]), k
        obs = np.random.RandomState(0).randn(7, 18)
        a0, _ = st["agent"].predict(obs, deterministic=True)
        a1, _ = again.predict(obs, deterministic=True)
        assert torch.equal(torch.as_tensor(a0).cpu(), torch.as_tensor(a1).cpu())


@pytest.mark.parametrize("cl", [["128", "128"], ["48", "32", "24"]])
def test_icrl_hc_wide_constraint_net_two_iterations_vs_port(golden, cl):
    """`-cl 128 128`: a constraint net with layers above 64 units — its cost inside per-step rollouts (the fused rollout's cost wave holds
    64 units), cost_function and train() 64 rows per workgroup with the weights in device memory — through the unchanged outer loop.
    `-cl 48 32 24`: more than two hidden layers (create_mlp takes any depth, torch_layers.py:93-126), same route."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", *cl, "-bi", "10", "-ft", "2000",
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.01", "-aclr", "0.9", "-crc", "0.5", "-psis",
            "-ctkno", "2.5", "-nt", "8", "--n_steps", "128", "-s", "3", "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1, streams=SeededStreams(13))
    st = setup(types.SimpleNamespace(**cfg))
    assert st["constraint_net"].wide and st["agent"]._fused_chain() is not None
    init = dict(policy={k: v.numpy().copy() for k, v in st["agent"].policy.state_dict().items()},
                cn={k: v.numpy().copy() for k, v in st["constraint_net"].state_dict().items()})
    ex = golden("expert_hc")
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"][:1000], ex["actions"][:1000], _sub(ex, "policy/"),
                                          streams=SeededStreams(13), init=init)
    keys = sorted(k for k in om[0] if k not in ("forward/std",))
    for it in range(2):
        m = outer_iteration(st, it)
        _compare(it, m, om[it], [k for k in keys if k in m], False, 2000)
    assert st["timesteps"] == steps == 2 * 2048


def test_icrl_antwall_two_iterations_vs_port(golden):
    """BASELINE configs[2] (AntWall ICRL, the reference's README.md:50 flags: constraint net [40, 40], batch 128 = two 64-row
    chunks -> two workgroups per network in the update kernel, clip_range 0.4, lambdas 0.9, lr 3e-5, nu0 0.1) on N = 8 envs,
    T = 128: HIP icrl() vs the CPU port over 2 outer iterations with the same noise / permutation streams."""
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    expert = os.path.join(HERE, "golden/expert_ant.npz")
    argv = ["icrl", "-ep", expert, "--expert_agent_path", expert, "-er", "2", "-cl", "40", "40", "-clr", "0.005", "-aclr", "0.9", "-crc", "0.6",
            "-bi", "5", "-ft", "2000", "-ni", "2", "-tei", "AntWall-v0", "-eei", "AntWallTest-v0", "--batch_size", "128",
            "--reward_gae_lambda", "0.9", "--cost_gae_lambda", "0.9", "--n_epochs", "4", "--learning_rate", "3e-5", "--clip_range", "0.4",
            "-piv", "0.1", "-plr", "0.05", "-psis", "-tk", "0.02", "-ctkno", "2.5", "-nt", "8", "--n_steps", "128", "-s", "5", "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1, streams=SeededStreams(23))
    st = setup(types.SimpleNamespace(**cfg))
    init = dict(policy={k: v.numpy().copy() for k, v in st["agent"].policy.state_dict().items()},
                cn={k: v.numpy().copy() for k, v in st["constraint_net"].state_dict().items()})
    ex = golden("expert_ant")
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, steps, _, objs = o_loop.icrl_port(port_cfg, ex["observations"][:1000], ex["actions"][:1000], _sub(ex, "policy/"),
                                          streams=SeededStreams(23), init=init)
    keys = sorted(k for k in om[0] if k not in ("forward/std",))
    worst = {}
    for it in range(2):
        m = outer_iteration(st, it)
        for k, v in _compare(it, m, om[it], [k for k in keys if k in m], False, 1000).items():
            worst[k] = max(worst.get(k, 0.0), v)
        assert abs(m["forward/std"] - om[it]["forward/std"]) < 1e-5
    assert st["timesteps"] == steps == 2 * 2048
    print("worst absolute deviation from the CPU port over 2 outer iterations (AntWall):",
          {k: float(f"{v:.3g}") for k, v in sorted(worst.items()) if v > 0})


@pytest.mark.parametrize("env_id,N,T,B,epochs", [("HCWithPos-v0", 256, 16, 64, 2), ("AntWallBroken-v0", 512, 8, 128, 2)])
def test_learn_at_per_gpu_shard_shapes_vs_port(env_id, N, T, B, epochs):
    """BASELINE configs[3] / configs[4] at the shape ONE GPU sees (2048 / 8 = 256 HC envs; 4096 / 8 = 512 AntWallBroken envs with
    the reference's frozen AntBroken constraint net, batch 128, clip 0.4, lr 3e-5, README.md:78): two rollouts + two
    PPO-Lagrangian updates through learn() against the CPU port on the same streams.  Tolerances as in the module docstring."""
    from icrl_amd import logger, utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    from oracle import nets as o_nets
    broken = "Broken" in env_id
    kind = "ant" if broken else "hc"
    od, ad = (113, 8) if broken else (18, 6)
    kw = dict(n_steps=T, batch_size=B, n_epochs=epochs, target_kl=0.01, seed=4, penalty_learning_rate=1.0 if broken else 0.1)
    if broken:
        kw.update(learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9)
    env = utils.make_train_env(env_id, None, True, 4, N, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    if broken:
        cn = ConstraintNet.load(os.path.join(HERE, "golden/cn_antbroken.npz"))
        ocn = o_nets.CostNet(od, ad, [40, 40], False, None, None, None, None, None)
    else:
        lo = -np.ones(ad, np.float32)
        torch.manual_seed(2)
        cn = ConstraintNet(od, ad, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
        ocn = o_nets.CostNet(od, ad, [20], False, None, None, 20, lo, -lo)
    ocn.load_state_dict(cn.state_dict())
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, streams=SeededStreams(21), **kw)
    stack = o_loop.make_stack(N, kind, 4, broken=broken); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, **kw)
    port.policy.load_state_dict(agent.policy.state_dict())
    agent.learn(2 * N * T)
    port.learn(2 * N * T, streams=SeededStreams(21))
    lg = dict(logger.Logger.CURRENT.name_to_value)
    assert agent.num_timesteps == port.num_timesteps == 2 * N * T
    assert abs(lg["train/nu"] - port.logs["train/nu"]) <= 1e-5
    assert lg["train/early_stop_epoch"] == port.logs["train/early_stop_epoch"]
    for k in ("train/average_cost", "train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss",
              "train/mean_reward_advantages", "train/mean_cost_advantages", "train/std"):
        assert _close(lg[k], port.logs[k], 2e-4, 2e-5), (k, lg[k], port.logs[k])
    n_steps = agent.policy.adam_step
    worst = 0.0
    for k, v in agent.policy.state_dict().items():
        ref = port.policy.params[k].detach().numpy()
        worst = max(worst, float(np.abs(v.numpy() - ref).max()))
        assert np.allclose(v.numpy(), ref, rtol=1e-3, atol=2e-5), (k, np.abs(v.numpy() - ref).max())
    print(f"{env_id} x {N}: {n_steps} optimiser steps, worst parameter deviation {worst:.3g}")


def _tol(k, ref, discrete=False, n_nominal=10000):
    """the module's per-key tolerance as one number (what _compare accepts around `ref`)."""
    if k in EXACT_INT:
        return 0.0
    if k == "forward/nu":
        return 1e-5
    if k in ("true/cost", "best_true/best_cost", "true/samples_behind", "true/samples_infront"):
        return 1.0 / n_nominal + 1e-12
    if k in ("true/reward", "true/reward_std", "best_true/best_reward"):
        return 1e-4 + 1e-4 * abs(ref)
    if k.startswith("backward/") or k in ("forward/approx_kl", "forward/clip_fraction"):
        return 2e-4 + 2e-3 * abs(ref)
    if k.endswith("explained_variance"):
        return 2e-5 + 1e-4 * abs(ref)
    return 1e-5 + 1e-4 * abs(ref)


@pytest.mark.parametrize("name", ["g19_whole_run_hc", "g20_whole_run_lgw", "g21_whole_run_ant"])
def test_icrl_whole_run_vs_port_band(golden, name):
    """north_star's result criterion over a WHOLE run (VERDICT r5 missing #2), for every single-GPU BASELINE config at FULL size:
      g19  configs[1]: HCWithPos-v0, 64 envs x 2048 steps, README.md:38 flags, 10 outer iterations = 2.6 M env steps, 20 rollouts + 20 updates of up to
           20 480 optimiser steps, 10 sampling / constraint-net / evaluation phases;
      g20  configs[0]: LGW-v0 / CLGW-v0, 1 env, n_steps 2000, README.md:25 flags, 5 outer iterations of 25 rollouts + updates (Categorical policy; the
           action uniforms are real uniforms here: SeededStreams(uniform=True));
      g21  configs[2]: AntWall-v0, 256 envs x 2048 steps, README.md:50 flags (batch 128, 20 epochs, constraint net [40, 40], 45 expert / nominal rollouts),
           6 outer iterations = 3.1 M env steps
    against tests/golden/<name>.npz: what the CPU port (pinned to the reference's own icrl() by g8) logged on the same SeededStreams and initial weights,
    undisturbed (`base`) and in 7 runs with a rounding-size disturbance (every initial parameter moved by -1 / 0 / +1 float32 ulp, the rows of every minibatch
    reversed / rotated, and — g20 / g21, where single discrete events carry the drift — a 1e-6-relative error in every tanh: tools/gen_whole_run.py).  Two
    correct fp32 executions of the algorithm separate over 10^5 dependent optimiser steps (LapGridWorld within ONE outer iteration), so the record is a
    BAND per metric and iteration: every metric of every iteration must lie in [lo - w - tol, hi + w + tol], w = hi - lo of the 8 port runs (a ninth sample
    of the same process), tol = the module's per-key tolerance (ref: icrl/icrl.py:199-304)."""
    import sys
    from icrl_amd.icrl import build_parser, setup, outer_iteration
    g = golden(name)
    spec = str(g["expert"]) if "expert" in g.files else "tests/golden/expert_hc.npz"
    if spec == "antwall45":
        sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
        import gen_whole_run
        expert = gen_whole_run.antwall45_expert(os.path.join("/tmp", f"icrl_whole_run_expert_ant45_{os.getpid()}.npz"))
    else:
        expert = os.path.join(os.path.dirname(HERE), spec)
    argv = [str(a) for a in g["argv"]] + ["-ep", expert, "--expert_agent_path", expert]
    cfg = vars(build_parser().parse_args(argv))
    uniform = bool(g["uniform_streams"]) if "uniform_streams" in g.files else False
    cfg.update(rank=0, world_size=1, streams=SeededStreams(int(g["stream_seed"]), uniform=uniform))
    st = setup(types.SimpleNamespace(**cfg))
    st["agent"].policy.load_state_dict(_sub(g, "w0/"))
    st["constraint_net"].load_state_dict(_sub(g, "cn0/"))
    keys = [str(k) for k in g["metric_keys"]]
    base, lo, hi = g["base"], g["lo"], g["hi"]
    n_it = base.shape[0]
    discrete = cfg["train_env_id"] == "LGW-v0"
    n_nominal = {"HCWithPos-v0": 10000, "LGW-v0": 4000, "AntWall-v0": 22500}[cfg["train_env_id"]]
    assert n_it >= 5 and {"forward/nu", "forward/average_cost", "true/cost", "true/reward", "backward/kl_new_old", "backward/kl_old_new"} <= set(keys)
    outside, table = [], []
    for it in range(n_it):
        m = outer_iteration(st, it)
        missing = [k for k in keys if k not in m]
        assert not missing, missing
        for j, k in enumerate(keys):
            x, b, l, h = float(m[k]), float(base[it, j]), float(lo[it, j]), float(hi[it, j])
            if k.endswith("explained_variance") and max(x, b, l, h) < 1.0:
                # 1 - Var[values - returns] / Var[values] is an ill-conditioned ratio when the critic is nearly constant (LapGridWorld: -28 .. -2e5 in the
                # port's own runs, -8e7 on the GPU): compared as Var[values] / Var[values - returns] = 1 / (1 - ev), monotone in ev and bounded
                x, b, l, h = (1.0 / (1.0 - v) for v in (x, b, l, h))
            if np.isnan(b) or np.isinf(b) or np.isnan(l) or np.isinf(l) or np.isinf(h):
                ok = True if not (np.isnan(b) or np.isinf(b)) else ((np.isnan(x) and np.isnan(b)) or x == b)      # (a band with a non-finite edge binds nothing)
            else:
                w, tol = h - l, _tol(k, b, discrete, n_nominal)
                ok = l - w - tol <= x <= h + w + tol
            if not ok:
                outside.append((it, k, x, b, l, h))
        j = keys.index
        table.append((it, m["forward/nu"], lo[it, j("forward/nu")], hi[it, j("forward/nu")], m["forward/average_cost"], lo[it, j("forward/average_cost")],
                      hi[it, j("forward/average_cost")], m["true/reward"], lo[it, j("true/reward")], hi[it, j("true/reward")], m["true/cost"],
                      lo[it, j("true/cost")], hi[it, j("true/cost")]))
    print(f"whole run ({name}), HIP vs the port's band [lo, hi] per outer iteration:")
    for r in table:
        print(f"  it {r[0]}: nu {r[1]:.6f} [{r[2]:.6f}, {r[3]:.6f}]  average_cost {r[4]:.5f} [{r[5]:.5f}, {r[6]:.5f}]  true/reward {r[7]:.1f} [{r[8]:.1f}, {r[9]:.1f}]  "
              f"true/cost {r[10]:.4f} [{r[11]:.4f}, {r[12]:.4f}]")
    per_it = {"HCWithPos-v0": 2 * 64 * 2048, "LGW-v0": 25 * 2000, "AntWall-v0": 256 * 2048}[cfg["train_env_id"]]
    assert st["timesteps"] == n_it * per_it
    assert not outside, outside[:20]
