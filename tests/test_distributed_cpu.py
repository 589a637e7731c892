"""CPU, world_size 2 over gloo: the single per-iteration all-reduce averages parameters and merges the running moments
EXACTLY (the merged statistics equal those of one process that saw both shards' streams)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HostRms:
    def __init__(self, n):
        self.mean, self.var, self.count = np.zeros(n), np.ones(n), 1e-4

    def assign(self, mean, var, count):
        self.mean, self.var, self.count = np.atleast_1d(np.asarray(mean, np.float64)), np.atleast_1d(np.asarray(var, np.float64)), float(count)

    def update(self, x):
        from oracle.stats import Moments
        m = Moments((len(self.mean),)); m.mean, m.var, m.count = self.mean.copy(), self.var.copy(), self.count
        m.update(x)
        self.mean, self.var, self.count = m.mean, m.var, m.count


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from icrl_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(100 + rank)
    params = torch.full((7,), float(rank + 1))
    moments = torch.arange(5, dtype=torch.float32) * (rank + 1)
    rms = HostRms(3)
    common = np.random.RandomState(7).randn(40, 3)
    rms.update(common)                                   # shared history
    prev = [D.moments_to_sums(rms.mean, rms.var, rms.count)]
    shard = rng.randn(25 + 10 * rank, 3) * (1 + rank) + rank
    rms.update(shard)                                    # this rank's own stream
    import types
    dual = types.SimpleNamespace(log_nu=np.float32(0.5 + rank), m=np.float32(0.1 * rank), v=np.float32(0.2), t=3)
    pol = types.SimpleNamespace(adam_step=100 + 30 * rank)          # ranks early-stopped at different epochs
    scal = D.Scalars(avg=[(dual, "log_nu"), (dual, "m"), (dual, "v")], counters=[(pol, "adam_step"), (dual, "t")])
    D.allreduce_state([params, moments], [rms], prev, world, scalars=scal)
    out[rank] = (params.numpy().copy(), moments.numpy().copy(), rms.mean.copy(), rms.var.copy(), rms.count, shard,
                 (float(dual.log_nu), float(dual.m), float(dual.v), dual.t, pol.adam_step, type(dual.log_nu).__name__))
    dist.destroy_process_group()


def test_allreduce_state_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    p0, m0, mean0, var0, c0, sh0, d0 = out[0]
    p1, m1, mean1, var1, c1, sh1, d1 = out[1]
    # dual variable averaged (stays float32), step counters agreed on: round(mean)
    assert d0 == d1 and d0[:3] == (1.0, float(np.float32(0.05)), float(np.float32(0.2))) and d0[3:] == (3, 115, "float32")
    assert np.allclose(p0, 1.5) and np.array_equal(p0, p1)                   # average of 1 and 2
    assert np.allclose(m0, np.arange(5) * 1.5) and np.array_equal(m0, m1)
    assert np.array_equal(mean0, mean1) and np.array_equal(var0, var1) and c0 == c1
    # ground truth: one stream that saw the common history and then both shards
    ref = HostRms(3)
    ref.update(np.random.RandomState(7).randn(40, 3)); ref.update(sh0); ref.update(sh1)
    assert np.allclose(mean0, ref.mean, rtol=1e-12, atol=1e-12)
    assert np.allclose(var0, ref.var, rtol=1e-10, atol=1e-12)
    assert abs(c0 - ref.count) < 1e-9


def test_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    from icrl_amd import distributed as D
    rms = HostRms(2); rms.update(np.random.RandomState(0).randn(10, 2))
    before = (rms.mean.copy(), rms.var.copy(), rms.count)
    t = torch.arange(4.0)
    D.allreduce_state([t], [rms], [D.moments_to_sums(*before)], 1)
    assert torch.equal(t, torch.arange(4.0)) and np.allclose(rms.mean, before[0]) and np.allclose(rms.var, before[1])


def _worker8(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from icrl_amd import distributed as D
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rms = HostRms(4)
    common = np.random.RandomState(11).randn(500, 4) * 3 + 50          # a long shared history, mean far from 0: the merge subtracts
    rms.update(common)                                                 # (G - 1) x its sums, so cancellation would show here
    rms.count = rms.count * 2000.0                                     # as if ~1e6 samples had been seen (configs[3]: 256 envs x 2048 rows x rollouts)
    prev = [D.moments_to_sums(rms.mean, rms.var, rms.count)]
    shard = np.random.RandomState(200 + rank).randn(131072 // 8, 4) * (1 + 0.1 * rank) + 50 + 0.01 * rank
    rms.update(shard)
    p = torch.full((3,), float(rank))
    D.allreduce_state([p], [rms], prev, world)
    out[rank] = (p.numpy().copy(), rms.mean.copy(), rms.var.copy(), rms.count, shard)
    dist.destroy_process_group()


def test_allreduce_state_world8_merge_is_exact_at_large_counts():
    """the shape the driver's 8-GPU run has: G = 8 streams sharing a history of ~1e6 samples; S_global = sum_g S_g - (G - 1) S_prev
    must reproduce the moments of ONE stream that saw the history and then all eight shards (variance to 1e-10 relative)."""
    G = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker8, args=(G, port, out), nprocs=G, join=True)
    ref = HostRms(4)
    ref.update(np.random.RandomState(11).randn(500, 4) * 3 + 50)
    ref.count = ref.count * 2000.0
    for r in range(G):
        ref.update(out[r][4])
    for r in range(G):
        p, mean, var, count, _ = out[r]
        assert np.allclose(p, np.mean(np.arange(G)))
        assert np.array_equal(mean, out[0][1]) and np.array_equal(var, out[0][2]) and count == out[0][3]      # identical on every rank
        assert abs(count - ref.count) <= 1e-9 * ref.count
        assert np.allclose(mean, ref.mean, rtol=1e-12, atol=0), np.abs(mean - ref.mean).max()
        assert np.allclose(var, ref.var, rtol=1e-10, atol=0), np.abs(var / ref.var - 1).max()
