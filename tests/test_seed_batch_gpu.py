"""GPU: several independent ICRL runs sharing one GPU inside the launches (icrl_amd/seed_batch.py: one host thread, every phase ONE
launch whose grid carries all runs, run = blockIdx.y; private random streams per run).  Every run must compute exactly what it
computes alone (bit-identical)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _cfg(seed, extra=(), envs=8, n_steps=64):
    from icrl_amd.icrl import build_parser
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "4", "-ft", str(2 * envs * n_steps),
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-crc", "0.5", "-psis", "-ctkno", "2.5",
            "-nt", str(envs), "--n_steps", str(n_steps), "-ne", "4", "-s", str(seed), "-v", "0", *extra]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1)
    return types.SimpleNamespace(**cfg)


def _snapshot(st, metrics):
    pol, cn, env = st["agent"].policy, st["constraint_net"], st["train_env"]
    return dict(params=pol.params.cpu().numpy().copy(), exp_avg_sq=pol.exp_avg_sq.cpu().numpy().copy(), cn=cn.params.cpu().numpy().copy(),
                obs_mean=np.asarray(env.obs_rms.mean).copy(), ret_var=float(env.ret_rms.var), nu=st["agent"].dual.nu().item(),
                rewards=st["agent"].rollout_buffer.rewards.cpu().numpy().copy(),
                metrics=[{k: v for k, v in m.items() if k not in ("time(m)", "time/fps", "time/time_elapsed")} for m in metrics])      # wall clock


def _solo(cfg, n_iters):
    """the run alone, through the ordinary single-run path (icrl.setup / icrl.outer_iteration) with the same private streams a
    batch gives it."""
    from icrl_amd import icrl as I, logger
    from icrl_amd.streams import PrivateStreams
    cfg.streams = PrivateStreams(cfg.seed)
    logger.configure()
    st = I.setup(cfg)
    return st, [I.outer_iteration(st, it) for it in range(n_iters)]


@pytest.mark.parametrize("extra,envs,n_steps", [((), 8, 64), (("-cbs", "64"), 8, 64), (("-rp",), 8, 64), ((), 16, 256)],
                         ids=["default", "cn_minibatches", "reset_policy", "16x256"])
def test_batched_runs_equal_solo_runs(extra, envs, n_steps):
    """default flags; constraint-net minibatch mode (-cbs: its permutations come from the run's own streams); --reset_policy (the
    agent is re-created, and the process-wide generators re-seeded, inside the loop); 16 envs x 256 steps (T x N x 4 bytes and the
    exchange workspace both large enough for the persistent rollout kernel)."""
    from icrl_amd.seed_batch import run_seed_batch
    seeds = [0, 1, 2, 3]
    solo = []
    for sd in seeds:
        st, m = _solo(_cfg(sd, extra, envs, n_steps), 2)
        solo.append(_snapshot(st, m))
    states, metrics, dt = run_seed_batch([_cfg(sd, extra, envs, n_steps) for sd in seeds], 2)
    assert len({s["nu"] for s in solo}) == len(seeds)            # the runs really are different runs
    for i, sd in enumerate(seeds):
        got = _snapshot(states[i], metrics[i])
        for k in ("params", "exp_avg_sq", "cn", "obs_mean", "rewards"):
            assert np.array_equal(got[k], solo[i][k]), (sd, k)
        assert got["ret_var"] == solo[i]["ret_var"] and got["nu"] == solo[i]["nu"]
        for a, b in zip(got["metrics"], solo[i]["metrics"]):
            assert a.keys() == b.keys()
            for k in a:
                assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), (sd, k, a[k], b[k])




def test_batched_entry_points_refuse_mixed_shapes():
    """the runs of a batch share every grid: a run with a different env count is refused by the host class, and by the C ABI."""
    from icrl_amd.seed_batch import SeedBatch
    with pytest.raises(ValueError, match="num_threads"):
        SeedBatch([_cfg(0), _cfg(1, envs=4)])
