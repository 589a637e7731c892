"""GPU parity of the persistent PPO-Lagrangian update kernel (icrl_ppo_lag_train) vs the reference (golden g4) and the
oracle's epoch loop (teacher-forced permutations).  fp32 tolerance: the kernel's MFMA dot products, tanh/exp and Adam
differ from torch-CPU by rounding order; after k dependent Adam steps parameters agree to ~1e-5 absolute (lr 3e-4)."""
import os

import numpy as np
import pytest
import torch

from oracle import nets as o_nets, ppo as o_ppo

pytestmark = pytest.mark.gpu


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


def _agent(kind, N, T, **kw):
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
    return PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=0, **kw)


def _fill(agent, buf):
    rb = agent.rollout_buffer
    for k, v in buf.items():
        getattr(rb, k).copy_(torch.as_tensor(np.asarray(v, np.float32)).reshape(getattr(rb, k).shape))
    rb.full = True


G13_ARCH = dict(pi=[32, 48], vf=[64, 32], cvf=[16, 64])          # oracle/gen_golden.py: g13_widths
G15_ARCH = dict(pi=[128, 128], vf=[96, 128], cvf=[128, 80])      # oracle/gen_golden.py: g15_wide (generic-shape path)
G17_ARCH = [48, dict(pi=[64, 32, 32], vf=[40], cvf=[])]          # oracle/gen_golden.py: g17_trunk (generic-shape path, `arch` descriptor)
G18_ARCH = [dict(pi=[32], vf=[64, 64, 64], cvf=[96, 200, 64, 16])]      # oracle/gen_golden.py: g18_deep


@pytest.mark.parametrize("name,arch", [("g4_ppo_minibatch", None), ("g13_widths", G13_ARCH), ("g14_batch256", None),
                                       ("g15_wide", G15_ARCH), ("g16_batch512", None), ("g17_trunk", G17_ARCH), ("g18_deep", G18_ARCH)])
def test_three_steps_on_one_batch_golden(golden, name, arch):
    """tests/golden/g4: the reference's own policy/optimizer objects stepped 3x on one 64-row batch.  g13: the same with
    -pl 32 48 -rvl 64 32 -cvl 16 64 (narrower layers run zero-padded to the kernels' 64; state dicts keep the logical shapes).
    g14: one 256-row batch (four chunks per minibatch).  g15: -pl 128 128 -rvl 96 128 -cvl 128 80 and g16: one 512-row batch — the
    shapes the persistent kernels do not serve run on the generic-shape path (csrc/generic.hip; icrl/utils.py:636-655,
    common/buffers.py:594-612).  g17: -sl 48 -pl 64 32 32 -rvl 40 -cvl (shared trunk, 3 / 1 / 0 layers) and g18: -pl 32 -rvl 64 64 64
    -cvl 96 200 64 16 — architectures described by icrl_policy_t.arch (torch_layers.py:129-254)."""
    g = golden(name)
    B = int(g["obs"].shape[0])
    kw = {} if arch is None else dict(policy_kwargs=dict(net_arch=[dict(arch)] if isinstance(arch, dict) else arch))
    agent = _agent("hc", 1, B, batch_size=B, n_epochs=3, target_kl=None, learning_rate=float(g["lr"]), **kw)
    agent.policy.load_state_dict(_sub(g, "w0/"))
    assert {k: tuple(v.shape) for k, v in agent.policy.state_dict().items()} == {k: g["w0/" + k].shape for k in agent.policy.shapes}
    z = np.zeros(B, np.float32)
    _fill(agent, dict(observations=g["obs"], actions=g["act"], log_probs=g["old_lp"], reward_advantages=g["adv_r"],
                      cost_advantages=g["adv_c"], reward_returns=g["ret_r"], cost_returns=g["ret_c"], reward_values=z,
                      cost_values=z, orig_costs=z))
    agent.dual.log_nu = np.float32(np.log(np.exp(float(g["nu"])) - 1))       # softplus^-1(nu)
    assert abs(agent.dual.nu().item() - float(g["nu"])) < 1e-6
    v_r, v_c, lp, _ = agent.policy.evaluate_actions(g["obs"], g["act"])        # the rollout-side forward on the same weights
    assert np.allclose(v_r.cpu().numpy().ravel(), g["s0/v_r"], rtol=1e-5, atol=2e-6)
    assert np.allclose(v_c.cpu().numpy().ravel(), g["s0/v_c"], rtol=1e-5, atol=2e-6)
    assert np.allclose(lp.cpu().numpy(), g["s0/log_prob"], rtol=1e-5, atol=2e-5)
    ident = np.tile(np.arange(B), (3, 1))
    agent.train(perms=ident)
    sd = agent.policy.state_dict()
    worst = 0.0
    for k, v in sd.items():
        ref = g["s2/after/" + k]
        worst = max(worst, float(np.abs(v.numpy() - ref).max()))
        assert np.allclose(v.numpy(), ref, rtol=1e-4, atol=3e-6), (k, np.abs(v.numpy() - ref).max())
    from icrl_amd import logger
    lg = logger.Logger.CURRENT.name_to_value
    assert lg["train/early_stop_epoch"] == 3
    ref_pg = np.mean([g[f"s{s}/policy_loss"] for s in range(3)])
    assert abs(lg["train/policy_gradient_loss"] - ref_pg) < 1e-5 + 1e-4 * abs(ref_pg)
    assert abs(lg["train/reward_value_loss"] - np.mean([g[f"s{s}/rvl"] for s in range(3)])) < 1e-4
    assert abs(lg["train/approx_kl"] - float(g["s2/approx_kl"])) < 1e-5
    assert abs(lg["train/clip_fraction"] - np.mean([g[f"s{s}/clip_fraction"] for s in range(3)])) < 1e-6
    assert agent.policy.adam_step == 3
    if arch is not None:           # the padding stayed exactly zero (weights and both Adam moments)
        pol = agent.policy
        for flat in (pol.params, pol.exp_avg, pol.exp_avg_sq):
            off, flat = 0, flat.cpu()
            for k, shp in pol.shapes.items():
                n = int(np.prod(shp))
                full = flat[off:off + n].reshape(shp).clone()
                full[tuple(slice(0, m) for m in pol.logical_shapes[k])] = 0
                assert float(full.abs().max()) == 0.0, k
                off += n
        osd = pol.optimizer_state_dict(lr=3e-4)
        assert tuple(osd["state"][1]["exp_avg"].shape) == g["w0/" + list(pol.shapes)[1]].shape      # (logical shape of the first weight)


ADAM_DEV_BOUND = 5e-4      # 3 x the largest deviation measured on MI355X (1.7e-4 x lr x steps; absolute: <= 9e-8, ~1 ulp)


def _oracle_train(agent_sd, buf, perms, kind, nu, okw=None, **h):
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    pol = o_nets.TwoCriticPolicy(od, ad, **(okw or {}))
    pol.load_state_dict(agent_sd)
    opt = torch.optim.Adam(pol.parameters(), lr=h.pop("lr"), eps=1e-5)
    out = o_ppo.ppo_lag_train(pol, opt, buf, perms, nu, **h)
    return pol, out


@pytest.mark.parametrize("kind,N,T,B,E,tk", [("hc", 8, 32, 64, 3, None), ("hc", 5, 40, 64, 2, None), ("ant", 24, 16, 128, 2, None),
                                             ("hc", 16, 32, 64, 6, 0.002), ("hc", 4, 8, 16, 2, None),
                                             ("hc", 5, 40, 128, 2, None),      # 200 rows: minibatches of 128 and 72 = chunks 64 + 64, 64 + 8
                                             ("ant", 3, 50, 100, 2, None),     # 150 rows: minibatches of 100 and 50 = chunks 64 + 36, 50
                                             # batch sizes above 128 (buffers.py:594-612 slices any size): one workgroup per network walks
                                             # up to four 64-row chunks; advantage statistics over up to 256 rows
                                             ("hc", 8, 64, 256, 2, None),      # 512 rows: minibatches of 256 = 4 chunks (wave pairs)
                                             ("hc", 5, 60, 200, 2, None),      # 300 rows: 200 (64 + 64 + 64 + 8) and 100 (64 + 36)
                                             ("ant", 4, 80, 256, 2, None),     # 320 rows: 256 and 64 (row-owning waves, obs 113)
                                             ("ant", 3, 70, 160, 2, None),     # 210 rows: 160 (3 chunks) and 50
                                             # obs 113 with minibatches of one chunk: four workgroups per network walking chunk by chunk (ppo_train_quarters.hip)
                                             ("ant", 6, 32, 64, 2, None), ("ant", 5, 20, 40, 2, None),
                                             ("ant", 16, 32, 128, 6, 0.002),   # target-KL stop through the four-workgroup form (the stop flag rides on a norm granule)
                                             # many steps at > 2 chunks per step: the schedule tables (16 B per step + 8 B per chunk)
                                             # must not reach the permutation offsets behind them (ICRL_PPO_PLAN_BYTES)
                                             ("hc", 8, 512, 256, 2, None),     # 32 steps x 4 chunks
                                             ("hc", 8, 256, 192, 3, None),     # 33 steps x 3 chunks
                                             # generic-shape path: layers wider than 64 / minibatches above 256 rows
                                             ("hc-wide", 8, 32, 64, 3, None), ("ant-wide", 6, 40, 128, 2, None), ("hc-wide", 16, 32, 64, 6, 0.002),
                                             ("hc", 8, 128, 512, 3, None), ("hc", 5, 200, 300, 2, None),      # 1000 rows: 300 + 300 + 300 + 100
                                             # architectures beyond three two-layer branches (-sl trunk, other depths; torch_layers.py:129-254)
                                             ("hc-trunk", 8, 32, 64, 3, None), ("ant-deep", 6, 40, 128, 2, None), ("hc-trunk-only", 5, 40, 100, 2, None),
                                             ("hc-bare", 4, 30, 40, 2, None), ("hc-deep", 16, 32, 64, 6, 0.002), ("ant-trunk", 4, 100, 400, 2, None),
                                             # the persistent generic-shape update at the edges of its workgroup layouts: ten row tiles x three branch workgroups
                                             # (150 rows: the last tile partial), eleven tiles = one workgroup per tile, a trunk whose shares are exchanged at 144 rows
                                             ("hc-wide", 6, 50, 150, 2, None), ("hc-wide", 7, 50, 175, 2, None), ("hc-trunk", 9, 48, 144, 2, None),
                                             # target-KL stop decided inside an Adam launch of ~3 900 workgroups: the epoch's last step is applied in full
                                             ("hc-huge", 8, 16, 64, 3, 1e-7)])
def test_train_vs_oracle(kind, N, T, B, E, tk, one_workgroup_per_network=False, train_kernel=None):
    from helpers.arches import ARCHES, oracle_arch_kwargs
    rng = np.random.RandomState(N * T)
    kind, _, shape = kind.partition("-")
    net_arch = [dict(pi=[128, 96], vf=[80, 128], cvf=[128, 128])] if shape == "wide" else ARCHES.get(shape)
    wide = net_arch is not None
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    lr = 3e-4 if kind == "hc" else 3e-5
    akw = dict(policy_kwargs=dict(net_arch=net_arch)) if wide else {}
    agent = _agent(kind, N, T, batch_size=B, n_epochs=E, target_kl=tk, learning_rate=lr, clip_range=0.2, **akw)
    if one_workgroup_per_network:
        agent.train_kernel = "rows1"
    if train_kernel is not None:
        agent.train_kernel = train_kernel
    sd0 = agent.policy.state_dict()
    obs = rng.randn(T, N, od).astype(np.float32)
    # old log-probs from the current policy on sampled actions so that ratios start near 1 (as in a real rollout)
    okw = oracle_arch_kwargs(net_arch) if wide else {}
    op = o_nets.TwoCriticPolicy(od, ad, **okw); op.load_state_dict(sd0)
    with torch.no_grad():
        a, vr, vc, lp = op.forward(torch.as_tensor(obs.reshape(-1, od)))
    buf = dict(observations=obs, actions=a.numpy().reshape(T, N, ad), log_probs=lp.numpy().reshape(T, N),
               reward_values=vr.numpy().reshape(T, N), cost_values=vc.numpy().reshape(T, N),
               reward_advantages=rng.randn(T, N).astype(np.float32) * 2, cost_advantages=rng.rand(T, N).astype(np.float32),
               reward_returns=rng.randn(T, N).astype(np.float32), cost_returns=rng.rand(T, N).astype(np.float32),
               orig_costs=rng.rand(T, N).astype(np.float32))
    _fill(agent, buf)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    nu = agent.dual.nu().item()
    agent.train(perms=perms)
    pol, out = _oracle_train(sd0, buf, perms, kind, nu, lr=lr, batch_size=B, n_epochs=E, clip_range=0.2, target_kl=tk, okw=okw)
    from icrl_amd import logger
    lg = logger.Logger.CURRENT.name_to_value
    assert lg["train/early_stop_epoch"] == out["train/early_stop_epoch"]
    n_steps = agent.policy.adam_step
    worst = 0.0
    for k, v in agent.policy.state_dict().items():
        ref = pol.params[k].detach().numpy()
        worst = max(worst, float(np.abs(v.numpy() - ref).max()))
    # measured on MI355X (printed below): after up to 48 dependent Adam steps no parameter is further than 9e-8 (about one
    # float32 ulp) from the oracle's, at most 1.7e-4 of the distance a parameter can travel (lr per step).  The bound is 3x
    # that plus one ulp; a wrong bias correction or moment update shifts every step by O(lr) and lands 3 orders above it.
    print(f"ADAM_DEV {kind} B={B} steps={n_steps}: max |param - oracle| = {worst:.3g} = {worst / (lr * n_steps):.3g} x lr x steps")
    assert worst <= ADAM_DEV_BOUND * lr * n_steps + 2e-7, (worst, worst / (lr * n_steps))
    for key in ("train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss", "train/approx_kl",
                "train/clip_fraction", "train/entropy_loss", "train/loss"):
        assert abs(lg[key] - out[key]) < 2e-4 + 2e-3 * abs(out[key]), (key, lg[key], out[key])
    assert np.allclose(agent.epoch_kls[:len(out["epoch_kls"])], out["epoch_kls"], atol=2e-5)


@pytest.mark.parametrize("tk", [None, 1e-6])
def test_numpy_stream_position_after_train(tk):
    """The reference draws one np.random.permutation per EXECUTED epoch (buffers.py:596 inside the epoch loop,
    ppo_lag.py:203-263: the KL break leaves the loop before the next epoch's get()).  The product draws all epochs' permutations
    before its single launch and then puts the global generator where the reference's would be; minibatch order is the
    numpy stream's."""
    N, T, E = 8, 32, 4
    agent = _agent("hc", N, T, batch_size=64, n_epochs=E, target_kl=tk)
    rng = np.random.RandomState(3)
    obs = rng.randn(T, N, 18).astype(np.float32)
    op = o_nets.TwoCriticPolicy(18, 6); op.load_state_dict(agent.policy.state_dict())
    with torch.no_grad():
        a, vr, vc, lp = op.forward(torch.as_tensor(obs.reshape(-1, 18)))
    _fill(agent, dict(observations=obs, actions=a.numpy().reshape(T, N, 6), log_probs=lp.numpy().reshape(T, N),
                      reward_values=vr.numpy().reshape(T, N), cost_values=vc.numpy().reshape(T, N),
                      reward_advantages=rng.randn(T, N).astype(np.float32), cost_advantages=rng.rand(T, N).astype(np.float32),
                      reward_returns=rng.randn(T, N).astype(np.float32), cost_returns=rng.rand(T, N).astype(np.float32),
                      orig_costs=rng.rand(T, N).astype(np.float32)))
    np.random.seed(1234)
    agent.train()
    from icrl_amd import logger
    stop = int(logger.Logger.CURRENT.name_to_value["train/early_stop_epoch"])
    executed = min(stop + 1, E)
    assert (executed < E) == (tk is not None)
    after = np.random.randint(1 << 30)
    np.random.seed(1234)
    for _ in range(executed):
        np.random.permutation(N * T)
    assert after == np.random.randint(1 << 30)


@pytest.mark.parametrize("N,T,B", [(24, 16, 128), (3, 50, 100)])
def test_two_chunk_minibatches_on_one_workgroup_per_network(N, T, B):
    """AntWall shapes with batch_size > 64: by default two workgroups per network each compute one 64-row chunk and exchange
    partial gradients (covered by test_train_vs_oracle above); hp._pad bit 3 keeps the sequential two-chunk loop of a single
    workgroup — same tolerances against the oracle."""
    test_train_vs_oracle("ant", N, T, B, 2, None, one_workgroup_per_network=True)


@pytest.mark.parametrize("N,T,B,E,tk", [(24, 16, 128, 2, None), (3, 50, 100, 2, None), (4, 80, 256, 2, None), (6, 32, 64, 2, None), (16, 32, 128, 6, 0.002)])
def test_ant_update_kernel_behind_the_default(N, T, B, E, tk):
    """obs 65..128: the default is FOUR workgroups per network (round 6: ppo_train_quarters2.hip for minibatches of 65..128 rows, both chunks in one
    pass; ppo_train_quarters.hip chunk by chunk otherwise — test_train_vs_oracle above).  `train_kernel = "rows"` (hp._pad & 4) keeps the row-owning
    kernel: two workgroups per network at 65..128 rows (rounds 3-5's default, and what a batched launch of several runs still uses), one otherwise."""
    test_train_vs_oracle("ant", N, T, B, E, tk, train_kernel="rows")


def test_ant_chunk_by_chunk_form_at_128_rows():
    """ICRL_QUARTERS_PASSES=1 keeps ppo_train_quarters.hip (16 rows of each 64-row chunk, one chunk after the other) where ppo_train_quarters2.hip
    would run: the oracle cases with 100- and 128-row minibatches through it (child process: the switch is read once per process)."""
    import subprocess, sys
    env = dict(os.environ, ICRL_QUARTERS_PASSES="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_train_vs_oracle and (ant-24-16-128 or ant-3-50-100)",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "2 passed" in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("kernel", ["halves", "pairs"])
@pytest.mark.parametrize("N,T,B,E,tk", [(8, 32, 64, 3, None), (16, 32, 64, 6, 0.002), (5, 40, 128, 2, None), (5, 60, 200, 2, None), (4, 8, 16, 2, None)])
def test_hc_update_kernels_behind_the_default(kernel, N, T, B, E, tk):
    """obs <= 32: the default is the wave-quad kernel with FOUR workgroups per network (round 6; test_train_vs_oracle above); batches of 17 .. 40 runs
    get its two-workgroup form (`train_kernel = "halves"`, hp._pad & 32: round 5's default), larger ones the wave-pair kernel (`"pairs"`, hp._pad & 16:
    rounds 2-4) — the same oracle cases through both, incl. ragged and multi-chunk minibatches and a target-KL stop."""
    test_train_vs_oracle("hc", N, T, B, E, tk, train_kernel=kernel)


def test_run_major_layout_equals_packed_layout():
    """the persistent update kernels and the multi-env rollout put a run's workgroups on one XCD (1-D grid: b, b + 8, ...) and store
    their granules with workgroup scope when every workgroup of the run reports the same XCD; any other dispatch uses the run-major
    grid and agent-scope stores.  Placement and store scope are speed matters only: ICRL_NO_XCD_PACK=1 forces the second form, and
    everything an update and a rollout leave must be bit-identical (child process: the switch is read once per process)."""
    import subprocess, sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "xcd_layout_child.py")
    digests = []
    for extra in ({}, {"ICRL_NO_XCD_PACK": "1"}):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1]


@pytest.mark.parametrize("kind,N,T,B", [("hc", 64, 256, 64), ("ant", 64, 256, 128)])
def test_sync_placement_tuning_leaves_no_trace(kind, N, T, B):
    """the first train() times short updates at several positions of the exchange workspace (PPOLagrangian._tune_sync_placement)
    and restores parameters, moments and the step counter: the update that follows is bit-identical to an untuned agent's, and
    a position was chosen."""
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    rng = np.random.RandomState(3)
    obs = rng.randn(T, N, od).astype(np.float32)
    buf = dict(observations=obs, actions=rng.randn(T, N, ad).astype(np.float32) * 0.5, log_probs=rng.randn(T, N).astype(np.float32) - 6,
               reward_values=rng.randn(T, N).astype(np.float32), cost_values=rng.rand(T, N).astype(np.float32),
               reward_advantages=rng.randn(T, N).astype(np.float32) * 2, cost_advantages=rng.rand(T, N).astype(np.float32),
               reward_returns=rng.randn(T, N).astype(np.float32), cost_returns=rng.rand(T, N).astype(np.float32),
               orig_costs=rng.rand(T, N).astype(np.float32))
    perms = np.stack([np.random.RandomState(7 + e).permutation(N * T) for e in range(2)])
    out = []
    for tune in (True, False):
        a = _agent(kind, N, T, batch_size=B, n_epochs=2, target_kl=None)
        a.tune_sync_placement = tune
        _fill(a, buf)
        a.train(perms=perms)
        out.append((a.policy.params.cpu().numpy().copy(), a.policy.exp_avg_sq.cpu().numpy().copy(), a.policy.adam_step,
                    a._train_ws.get("sync_position")))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    assert out[0][3] in range(4) and out[1][3] is None


@pytest.mark.parametrize("switch", ["ICRL_GEN_LAUNCHES", "ICRL_GEN_BRANCH_WGS", "ICRL_NO_XCD_PACK"])
def test_generic_shape_update_as_one_persistent_launch(switch):
    """The generic-shape update is ONE cooperative launch per train() by default (csrc/generic.hip: one workgroup per (row tile, branch) on one
    XCD, fp32 MFMA tiles, three grid barriers per optimiser step).  Its other forms stay correct — the golden and oracle
    cases of the generic-shape path pass under each of them (child process: the switches are read once per process):
    ICRL_GEN_LAUNCHES=1 three plain launches per optimiser step (the fallback of shapes the persistent form does not hold),
    ICRL_GEN_BRANCH_WGS=1 one workgroup per row tile walking all three branches (batches above 160 rows take it anyway),
    ICRL_NO_XCD_PACK=1 the dense grid: workgroups on several XCDs, agent-scope stores instead of the L2-local exchange."""
    import subprocess, sys
    env = dict(os.environ, **{switch: "1"})
    sel = "g15 or g16 or g17 or g18 or hc-wide or hc-trunk or ant-deep or hc-bare or hc-8-128-512 or hc-5-200-300"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:]


@pytest.mark.parametrize("shape,B", [("wide", 64), ("trunk", 64), ("trunk", 144), ("deep", 320)])
def test_generic_shape_update_is_reproducible(shape, B):
    """The persistent generic-shape update sums across workgroups in fixed orders only (row tiles in tile order, the trunk's three shares in branch
    order, fixed trees inside a workgroup): two runs from the same state leave bit-identical parameters, Adam moments and logged sums — whatever
    order the workgroups reach the barriers in."""
    from helpers.arches import ARCHES
    from icrl_amd import logger
    net_arch = [dict(pi=[128, 96], vf=[80, 128], cvf=[128, 128])] if shape == "wide" else ARCHES[shape]
    N, T = 8, 160
    rng = np.random.RandomState(17)
    out = []
    for rep in range(3):
        agent = _agent("hc", N, T, batch_size=B, n_epochs=2, target_kl=None, learning_rate=3e-4, policy_kwargs=dict(net_arch=net_arch))
        if rep == 0:
            sd0 = {k: v.clone() for k, v in agent.policy.state_dict().items()}
            data = dict(observations=rng.randn(T, N, 18), actions=rng.randn(T, N, 6), log_probs=-3 + 0.1 * rng.randn(T, N), reward_advantages=rng.randn(T, N),
                        cost_advantages=rng.randn(T, N), reward_returns=rng.randn(T, N), cost_returns=rng.randn(T, N), reward_values=rng.randn(T, N),
                        cost_values=rng.randn(T, N), orig_costs=np.abs(rng.randn(T, N)))
            perms = np.stack([np.random.RandomState(3 + e).permutation(N * T) for e in range(2)])
        agent.policy.load_state_dict(sd0)
        _fill(agent, data)
        agent.train(perms=perms)
        lg = logger.Logger.CURRENT.name_to_value
        out.append(({k: v.clone() for k, v in agent.policy.state_dict().items()}, agent.policy.exp_avg.clone().cpu(), agent.policy.exp_avg_sq.clone().cpu(),
                    {k: float(v) for k, v in lg.items() if k.startswith("train/")}))
    for rep in (1, 2):
        for k in out[0][0]:
            assert torch.equal(out[0][0][k], out[rep][0][k]), (rep, k)
        assert torch.equal(out[0][1], out[rep][1]) and torch.equal(out[0][2], out[rep][2]) and out[0][3] == out[rep][3]
    assert any(float((out[0][0][k] - sd0[k]).abs().max()) > 0 for k in sd0)


@pytest.mark.parametrize("kind,B", [("hc", 64), ("ant", 128)])
def test_multi_workgroup_update_reports_a_workgroup_that_never_arrives(kind, B):
    """The four-workgroups-per-network update kernels (ppo_train_halves.hip at HC widths, ppo_train_quarters2.hip at AntWall's) exchange partial gradients and
    norm granules between twelve workgroups; every one of those waits is bounded.  With hp._pad & 64 the run's last workgroup leaves before the first
    optimiser step: the others give up at their flag / granule polls (seconds), the launch ENDS with the status word set and the host raises — no hung GPU."""
    import time
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    N, T = 8, 32
    agent = _agent(kind, N, T, batch_size=B, n_epochs=2, target_kl=None)
    rng = np.random.RandomState(2)
    _fill(agent, dict(observations=rng.randn(T, N, od), actions=rng.randn(T, N, ad), log_probs=-3 + 0.1 * rng.randn(T, N), reward_advantages=rng.randn(T, N),
                      cost_advantages=rng.randn(T, N), reward_returns=rng.randn(T, N), cost_returns=rng.randn(T, N), reward_values=rng.randn(T, N),
                      cost_values=rng.randn(T, N), orig_costs=np.abs(rng.randn(T, N))))
    agent.tune_sync_placement = False
    agent.profile_phases = 64
    t0 = time.time()
    with pytest.raises(RuntimeError, match="timed out"):
        agent.train()
    print(f"{kind}: the launch with a missing workgroup ended after {time.time() - t0:.1f} s")
    assert time.time() - t0 < 120
    agent.profile_phases = 0


def test_generic_shape_update_reports_a_workgroup_that_never_arrives():
    """Every wait of the persistent generic-shape update is bounded: with hp._pad & 64 the last workgroup leaves before the first optimiser step, the
    others give up at the first grid barrier and the launch ENDS (no hung GPU) with the status word set — the host raises like for the other persistent kernels."""
    from helpers.arches import ARCHES
    agent = _agent("hc", 8, 32, batch_size=64, n_epochs=2, target_kl=None, policy_kwargs=dict(net_arch=ARCHES["trunk"]))
    rng = np.random.RandomState(2)
    T, N = 32, 8
    _fill(agent, dict(observations=rng.randn(T, N, 18), actions=rng.randn(T, N, 6), log_probs=-3 + 0.1 * rng.randn(T, N), reward_advantages=rng.randn(T, N),
                      cost_advantages=rng.randn(T, N), reward_returns=rng.randn(T, N), cost_returns=rng.randn(T, N), reward_values=rng.randn(T, N),
                      cost_values=rng.randn(T, N), orig_costs=np.abs(rng.randn(T, N))))
    agent.profile_phases = 64
    with pytest.raises(RuntimeError, match="timed out"):
        agent.train()
    agent.profile_phases = 0
    agent.train()      # the next launch is sound


@pytest.mark.parametrize("tk", [None, 1e-6, 0.02])
def test_epochwise_update_equals_single_launch(tk):
    """PPOLagrangian._train_epochwise (rollouts of >= LAZY_PERM_ROWS rows: one launch per epoch, the next epoch's np.random.permutation
    drawn beside the running epoch; ref: ppo_lag.py:203-299, buffers.py:596) against the single launch on the same buffer and generator
    state: parameters, Adam moments and step counter bit-identical, the same epochs executed, np.random left where the reference
    leaves it, the logged means equal to float32 rounding of their sums."""
    from icrl_amd import logger
    N, T, E = 8, 64, 5
    rng = np.random.RandomState(5)
    obs = rng.randn(T, N, 18).astype(np.float32)
    out = []
    for lazy_rows in (1 << 30, 1):
        agent = _agent("hc", N, T, batch_size=64, n_epochs=E, target_kl=tk, learning_rate=1e-3)
        agent.LAZY_PERM_ROWS = lazy_rows
        op = o_nets.TwoCriticPolicy(18, 6); op.load_state_dict(agent.policy.state_dict())
        with torch.no_grad():
            a, vr, vc, lp = op.forward(torch.as_tensor(obs.reshape(-1, 18)))
        r2 = np.random.RandomState(6)
        _fill(agent, dict(observations=obs, actions=a.numpy().reshape(T, N, 6), log_probs=lp.numpy().reshape(T, N),
                          reward_values=vr.numpy().reshape(T, N), cost_values=vc.numpy().reshape(T, N),
                          reward_advantages=r2.randn(T, N).astype(np.float32), cost_advantages=r2.rand(T, N).astype(np.float32),
                          reward_returns=r2.randn(T, N).astype(np.float32), cost_returns=r2.rand(T, N).astype(np.float32),
                          orig_costs=r2.rand(T, N).astype(np.float32)))
        np.random.seed(4321)
        agent.train()
        lg = dict(logger.Logger.CURRENT.name_to_value)
        out.append(dict(params=agent.policy.params.clone(), m=agent.policy.exp_avg.clone(), v=agent.policy.exp_avg_sq.clone(), t=agent.policy.adam_step,
                        after=np.random.randint(1 << 30), lg=lg, kls=np.asarray(agent.epoch_kls).copy(), nu=agent.dual.nu().item()))
    a_, b_ = out
    assert torch.equal(a_["params"], b_["params"]) and torch.equal(a_["m"], b_["m"]) and torch.equal(a_["v"], b_["v"]) and a_["t"] == b_["t"] > 0
    assert a_["after"] == b_["after"] and a_["nu"] == b_["nu"]
    assert a_["lg"]["train/early_stop_epoch"] == b_["lg"]["train/early_stop_epoch"]
    if tk == 1e-6:
        assert a_["lg"]["train/early_stop_epoch"] < E - 1          # (the loop did end early: the epoch-wise form stopped launching)
    n_exec = int(min(a_["lg"]["train/early_stop_epoch"] + 1, E))
    assert np.array_equal(a_["kls"][:n_exec], b_["kls"][:n_exec])
    for key in ("train/approx_kl", "train/loss"):
        assert a_["lg"][key] == b_["lg"][key], key
    for key in ("train/entropy_loss", "train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss", "train/clip_fraction"):
        assert abs(a_["lg"][key] - b_["lg"][key]) <= 1e-6 * max(1.0, abs(a_["lg"][key])), (key, a_["lg"][key], b_["lg"][key])
