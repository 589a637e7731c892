"""Multi-GPU: one process per GPU, env shards, ONE all-reduce per outer ICRL iteration (RCCL over xGMI via torch.distributed
backend "nccl"; "gloo" on CPU for the tests).

The reference has no multi-device mode.  Rank r owns envs [r*N, (r+1)*N) (stream keys seed + global index),
runs its PPO epochs and its constraint-net update on its own shard, then everything that must stay common is averaged /
merged with a single flat float64 all-reduce (SUM):
    [ policy params | policy exp_avg | policy exp_avg_sq | cn params | cn exp_avg | cn exp_avg_sq | dual (log_nu, m, v) |
      step counters (policy Adam, cn Adam, dual Adam) | obs_rms S | ret_rms S | cost_rms S ]
with S = (count, count*mean, count*(var + mean^2)).
  * parameters / Adam moments / the dual variable are averaged (local-update data parallelism: the mean of the ranks'
    parameter deltas is applied once per outer iteration);
  * step counters become round(mean): ranks that stop a PPO epoch loop early at different epochs (target-KL) hold moments
    of different ages; one common counter keeps the bias corrections identical on every rank afterwards;
  * running moments merge exactly: S_global = sum_g S_g - (G-1) * S_previous_common.
At <= 0.5 MB the message is latency-bound (7 x 153 GB/s xGMI links are irrelevant), hence ONE call.  The flat buffer is
assembled and unpacked with device ops only (one torch.cat, no host read-back); with the gloo backend it takes one
round trip through host memory.
"""
import numpy as np
import torch
import torch.distributed as dist


def init_from_env():
    import os
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return 0, 1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    # ICRL_DIST_BACKEND=gloo: functional check of the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # device LOCAL_RANK % device_count; the reduction goes through host memory)
    backend = os.environ.get("ICRL_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return rank, world


def reduce_device(default="cuda"):
    """where collectives' tensors live: device memory for RCCL, host memory for gloo."""
    return "cpu" if (dist.is_initialized() and dist.get_backend() == "gloo") else default


def broadcast_seed(seed, rank, world):
    """one seed for the whole job: rank 0's (the reference draws np.random.randint(0, 100) when --seed is absent,
    icrl/icrl.py:433-434; every rank must then build the same initial networks)."""
    if world <= 1:
        return seed
    t = torch.tensor([-1 if seed is None else int(seed)], dtype=torch.int64, device=reduce_device())
    dist.broadcast(t, src=0)
    return int(t.item())


def decorrelate_streams(seed, rank):
    """after the (common) weight initialisation: give rank r its own action-noise and permutation streams.  Rank 0 keeps
    the single-GPU streams."""
    if rank == 0:
        return
    np.random.seed((int(seed) + 7919 * rank) % (2 ** 32))
    torch.manual_seed(int(seed) + 7919 * rank)       # seeds the CPU and every device generator


def moments_to_sums(mean, var, count):
    mean, var = np.asarray(mean, np.float64).reshape(-1), np.asarray(var, np.float64).reshape(-1)
    return np.concatenate([[count], count * mean, count * (var + mean * mean)])


def sums_to_moments(s, n):
    count = s[0]
    mean = s[1:1 + n] / count
    var = s[1 + n:1 + 2 * n] / count - mean * mean
    return mean, var, count


def merge_sums(all_reduced, previous_common, world):
    """exact merge of `world` streams that share the history `previous_common`."""
    return all_reduced - (world - 1) * previous_common


def _rms_sums(r, dev):
    """S of a running-moments object as a float64 tensor on `dev`; device-resident objects (vec_env.RunningMeanStd /
    _ScalarRms) are read with device ops, host objects (.mean / .var / .count) through numpy."""
    if hasattr(r, "d_stats"):
        m, v, c = r.d_stats[0:1], r.d_stats[1:2], r.d_stats[2:3]
    elif hasattr(r, "d_mean"):
        m, v, c = r.d_mean, r.d_var, r.d_count
    else:
        return torch.as_tensor(moments_to_sums(r.mean, r.var, r.count), dtype=torch.float64, device=dev)
    return torch.cat([c, c * m, c * (v + m * m)]).to(dev)


def _rms_assign(r, merged):
    n = (merged.numel() - 1) // 2
    count = merged[0:1]
    mean = merged[1:1 + n] / count
    var = merged[1 + n:1 + 2 * n] / count - mean * mean
    if hasattr(r, "d_stats"):
        r.d_stats.copy_(torch.cat([mean, var, count]).to(r.d_stats.device))
    elif hasattr(r, "d_mean"):
        r.d_mean.copy_(mean.to(r.d_mean.device)); r.d_var.copy_(var.to(r.d_var.device)); r.d_count.copy_(count.to(r.d_count.device))
    else:
        mean, var = mean.cpu().numpy(), var.cpu().numpy()
        r.assign(mean if n > 1 else mean[0], var if n > 1 else var[0], float(count.item()))


class Scalars:
    """host scalars that ride in the same message: `avg` (obj, attr) pairs are averaged, `counters` become round(mean)."""

    def __init__(self, avg=(), counters=()):
        self.avg, self.counters = list(avg), list(counters)

    def __len__(self):
        return len(self.avg) + len(self.counters)

    def pack(self, dev):
        vals = [float(getattr(o, a)) for o, a in self.avg + self.counters]
        return torch.tensor(vals, dtype=torch.float64, device=dev)

    def unpack(self, red, world):
        vals = red.cpu().numpy() / world
        for (o, a), v in zip(self.avg, vals[:len(self.avg)]):
            setattr(o, a, type(getattr(o, a))(v))
        for (o, a), v in zip(self.counters, vals[len(self.avg):]):
            setattr(o, a, int(np.rint(v)))


def allreduce_state(avg_tensors, rms_list, rms_prev_sums, world, group=None, scalars=None, force_collective=False):
    """avg_tensors: tensors to average in place; rms_list: running-moment objects (device-resident or host objects with
    .mean/.var/.count/.assign); rms_prev_sums: their S at the last synchronisation; scalars: optional Scalars.
    ONE all-reduce.  Returns the new common S list (float64 tensors).  force_collective: issue the all-reduce at world size 1 too
    (tests/test_multirank_gpu.py exercises the RCCL branch — device-side float64 SUM, no host round trip — on a 1-GPU box)."""
    dev = avg_tensors[0].device if avg_tensors else torch.device("cpu")
    sums = [_rms_sums(r, dev) for r in rms_list]
    parts = [t.detach().reshape(-1).double() for t in avg_tensors]
    if scalars is not None and len(scalars):
        parts.append(scalars.pack(dev))
    buf = torch.cat(parts + sums)
    if world > 1 or (force_collective and dist.is_initialized()):
        if buf.device.type == "cuda" and reduce_device() == "cpu":
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            buf = host.to(buf.device)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in avg_tensors:
        n = t.numel()
        t.copy_((buf[off:off + n] / world).reshape(t.shape).to(t.dtype))
        off += n
    if scalars is not None and len(scalars):
        scalars.unpack(buf[off:off + len(scalars)], world)
        off += len(scalars)
    new_sums = []
    for r, s, prev in zip(rms_list, sums, rms_prev_sums):
        n = s.numel()
        prev = torch.as_tensor(prev, dtype=torch.float64, device=dev)
        merged = buf[off:off + n] - (world - 1) * prev if world > 1 else buf[off:off + n]
        off += n
        _rms_assign(r, merged)
        new_sums.append(merged.clone())
    return new_sums
