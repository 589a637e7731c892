"""GPU: several independent ICRL runs sharing one GPU (icrl_amd/seed_batch.py: a stream + a host thread + private random streams
per run, admission control for the persistent launches).  Every run must compute exactly what it computes alone."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _cfg(seed):
    from icrl_amd.icrl import build_parser
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "4", "-ft", "1024",
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-crc", "0.5", "-psis", "-ctkno", "2.5",
            "-nt", "8", "--n_steps", "64", "-ne", "4", "-s", str(seed), "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1)
    return types.SimpleNamespace(**cfg)


def _snapshot(st, metrics):
    pol, cn, env = st["agent"].policy, st["constraint_net"], st["train_env"]
    return dict(params=pol.params.cpu().numpy().copy(), exp_avg_sq=pol.exp_avg_sq.cpu().numpy().copy(), cn=cn.params.cpu().numpy().copy(),
                obs_mean=np.asarray(env.obs_rms.mean).copy(), ret_var=float(env.ret_rms.var), nu=st["agent"].dual.nu().item(),
                rewards=st["agent"].rollout_buffer.rewards.cpu().numpy().copy(),
                metrics=[{k: v for k, v in m.items() if k != "time(m)"} for m in metrics])


def test_batched_runs_equal_solo_runs():
    from icrl_amd.seed_batch import run_seed_batch
    seeds = [0, 1, 2, 3]
    solo = []
    for sd in seeds:
        st, m, _ = run_seed_batch([_cfg(sd)], 2)
        solo.append(_snapshot(st[0], m[0]))
    states, metrics, dt = run_seed_batch([_cfg(sd) for sd in seeds], 2)
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


def test_cu_budget_serialises_oversubscription():
    """six 64-env rollouts want 384 CUs: the budget admits four at a time; nothing times out."""
    from icrl_amd import _lib
    from icrl_amd.seed_batch import CuBudget
    b = CuBudget(256)
    got = [b.acquire(64) for _ in range(4)]
    assert b.free == 0 and got == [64] * 4
    import threading
    done = []
    t = threading.Thread(target=lambda: done.append(b.acquire(64)))
    t.start(); t.join(0.2)
    assert t.is_alive() and not done                              # the fifth waits
    b.release(64); t.join(2.0)
    assert done == [64] and b.free == 0
    for _ in range(4):
        b.release(64)
    assert b.acquire(1000) == 256 and b.free == 0                 # a request larger than the chip is clamped, not deadlocked
