"""GPU: the multi-rank path as a working whole — two fresh child processes (one per rank, gloo transport, both on GPU 0;
RCCL itself needs >= 2 GPUs and is exercised by the driver's scaling run) run sharded rollouts and 2 outer ICRL iterations
with the per-iteration collective (icrl_amd/distributed.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def two_ranks(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("ranks")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ICRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "helpers/two_rank_child.py"), str(tmp / f"r{rank}.npz")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o[-3000:]}"
    return [np.load(tmp / f"r{rank}.npz") for rank in range(2)]


def test_shards_are_distinct_and_initial_networks_common(two_ranks):
    """re-seeding through the wrapper chain must keep rank r on envs [rN, (r+1)N)."""
    r0, r1 = two_ranks
    for ph, seed in (("A", 5), ("B", 9)):
        assert list(r0[f"{ph}_keys"]) == [seed + i for i in range(8)]
        assert list(r1[f"{ph}_keys"]) == [seed + 8 + i for i in range(8)]
        assert np.array_equal(r0[f"{ph}_params0"], r1[f"{ph}_params0"])
    assert not np.array_equal(r0["A_orig_obs"], r1["A_orig_obs"])
    assert not np.array_equal(r0["B_first_obs"], r1["B_first_obs"])


def test_sharded_rollout_equals_one_process_over_the_union(two_ranks):
    """without observation normalisation the policy input does not depend on the statistics, so the two shards together ARE
    the 16-env rollout of one process; after the collective every rank holds the running moments of that one process."""
    from icrl_amd import utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    sys.path.insert(0, os.path.join(HERE, "helpers"))
    import two_rank_child as C
    N, T, A = C.N, C.T, C.A
    env = utils.make_train_env("HCWithPos-v0", None, True, 5, 2 * N, normalize_obs=False, cost_info_str="cost", reward_gamma=0.99,
                               cost_gamma=0.99)
    lo = -np.ones(A, np.float32)
    torch.manual_seed(1)
    cn = ConstraintNet(18, A, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, n_epochs=2, seed=5)
    assert np.array_equal(agent.policy.params.cpu().numpy(), two_ranks[0]["A_params0"])
    noise = np.concatenate([C.shard_noise(3, 0, 0), C.shard_noise(3, 0, 1)], axis=1)
    agent._setup_learn(2 * N * T)
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=torch.as_tensor(noise, device="cuda"))
    oo = agent.rollout_buffer.new_orig_observations.cpu().numpy()
    for r in range(2):
        assert np.array_equal(oo[:, r * N:(r + 1) * N], two_ranks[r]["A_orig_obs"]), r        # float64 env streams: bit-exact
    for name, rms in (("obs", env.obs_rms), ("ret", env.ret_rms), ("cost", env.cost_rms)):
        one = np.concatenate([np.atleast_1d(rms.mean), np.atleast_1d(rms.var), [rms.count]])
        for r in range(2):
            got = two_ranks[r][f"A_{name}_rms"]
            # Chan merges in a different association order: equal to float64 rounding
            assert np.allclose(got, one, rtol=1e-10, atol=1e-13), (name, r, np.abs(got - one).max())
        assert np.array_equal(two_ranks[0][f"A_{name}_rms"], two_ranks[1][f"A_{name}_rms"])


def test_state_identical_across_ranks_after_two_outer_iterations(two_ranks):
    r0, r1 = two_ranks
    for k in ("B_params", "B_exp_avg", "B_exp_avg_sq", "B_cn", "B_cn_m", "B_cn_v", "B_dual", "B_steps", "B_obs_rms", "B_ret_rms",
              "B_cost_rms"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.all(np.isfinite(r0["B_params"])) and r0["B_steps"][0] > 0 and r0["B_dual"][3] == 4      # 2 x 2 train() calls
    assert r0["B_timesteps"][0] == 2 * 1024
    # the logged nu is each rank's own pre-collective value: shards differ, so these differ, while the carried state agrees
    assert not np.array_equal(r0["B_logged_nu"], r1["B_logged_nu"])
    # merged observation count = initial 1e-4 + both shards' samples: 2 iterations x (reset excluded) 2 rollouts x 64 x 16 envs
    assert abs(r0["B_obs_rms"][-1] - (1e-4 + 2 * 2 * 64 * 16)) < 1e-6


def test_cpg_ranks_agree_after_learn(two_ranks):
    """BASELINE configs[4]'s path (cpg, frozen constraint net) on two ranks: own env shards and noise, ONE all-reduce per rollout + update
    (callbacks.RankSyncCallback) and one when learn() ends — parameters, Adam moments, the dual variable, the step counters and the three
    running-moment sets are then identical on both ranks, and the merged observation count is both shards' samples."""
    r0, r1 = two_ranks
    assert list(r0["C_keys"]) == [4 + i for i in range(16)] and list(r1["C_keys"]) == [4 + 16 + i for i in range(16)]
    assert not np.array_equal(r0["C_first_obs"], r1["C_first_obs"])
    for k in ("C_params", "C_exp_avg", "C_exp_avg_sq", "C_dual", "C_steps", "C_obs_rms", "C_ret_rms", "C_cost_rms"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.all(np.isfinite(r0["C_params"])) and r0["C_steps"][0] > 0 and r0["C_dual"][3] == 3
    assert r0["C_timesteps"][0] == 3 * 16 * 64
    assert abs(r0["C_obs_rms"][-1] - (1e-4 + 3 * 64 * 32)) < 1e-6      # 3 rollouts x 64 steps x (16 + 16) envs (the reset is not counted)


def test_rccl_allreduce_state_one_rank(tmp_path):
    """the `nccl` (RCCL) branch of the per-iteration collective on real hardware: a fresh child process with a 1-rank nccl process
    group runs distributed.allreduce_state with the collective forced — RCCL loads, a float64 SUM of the ~33 k-element flat buffer runs
    on the device (no host round trip), and at world size 1 it is the identity."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "nccl1.npz"
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ICRL_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "helpers/nccl_one_rank_child.py"), str(out)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = np.load(out)
    assert str(r["seen_device"]) == "cuda" and str(r["seen_dtype"]) == "torch.float64"
    assert int(r["seen_numel"]) == 2 * 16654 + 5 + (1 + 2 * 18)
    assert bool(r["same_params"]) and bool(r["same_m"])
    assert float(r["mean_dev"]) < 1e-12 and float(r["var_dev"]) < 1e-10 and abs(float(r["count"]) - 1234.5) < 1e-9
    assert np.array_equal(r["x"], np.arange(8) + 0.125)
    assert np.allclose(r["scal"], [0.5, 0.1, 0.2, 3, 100])
    print("RCCL version:", r["rccl"])
