"""child of tests/test_multirank_gpu.py::test_rccl_allreduce_state_one_rank: a 1-rank `nccl` (= RCCL) process group on GPU 0
running distributed.allreduce_state's device branch with the collective forced.  Writes what it saw to argv[1]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(out):
    from icrl_amd import distributed as D
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl" and D.reduce_device() == "cuda"
    from icrl_amd.vec_env import RunningMeanStd
    g = torch.Generator(device="cuda").manual_seed(3)
    params = torch.randn(16654, device="cuda", generator=g)
    m = torch.randn(16654, device="cuda", generator=g)
    rms = RunningMeanStd(shape=(18,), device="cuda")
    rms.assign(np.arange(18) * 0.25, 1.0 + np.arange(18) * 0.5, 1234.5)
    prev = [D.moments_to_sums(rms.mean, rms.var, rms.count)]
    before = (params.clone(), m.clone(), rms.mean.copy(), rms.var.copy(), rms.count)
    # spy on the collective: the buffer must be a float64 CUDA tensor (no host round trip)
    seen = {}
    real = dist.all_reduce

    def spy(t, *a, **k):
        seen.update(device=t.device.type, dtype=str(t.dtype), numel=int(t.numel()))
        return real(t, *a, **k)
    dist.all_reduce = spy
    import types
    dual = types.SimpleNamespace(log_nu=np.float32(0.5), m=np.float32(0.1), v=np.float32(0.2), t=3)
    pol = types.SimpleNamespace(adam_step=100)
    scal = D.Scalars(avg=[(dual, "log_nu"), (dual, "m"), (dual, "v")], counters=[(pol, "adam_step"), (dual, "t")])
    D.allreduce_state([params, m], [rms], prev, 1, scalars=scal, force_collective=True)
    dist.all_reduce = real
    torch.cuda.synchronize()
    # a second, plain float64 SUM of known values
    x = torch.arange(8, dtype=torch.float64, device="cuda") + 0.125
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    np.savez(out, same_params=bool(torch.equal(params, before[0])), same_m=bool(torch.equal(m, before[1])),
             mean_dev=float(np.abs(rms.mean - before[2]).max()), var_dev=float(np.abs(rms.var - before[3]).max()), count=rms.count,
             seen_device=seen.get("device", ""), seen_dtype=seen.get("dtype", ""), seen_numel=seen.get("numel", 0),
             x=x.cpu().numpy(), scal=np.array([float(dual.log_nu), float(dual.m), float(dual.v), dual.t, pol.adam_step]),
             rccl=str(getattr(torch.cuda.nccl, "version", lambda: "?")()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
