"""Multi-GPU: one process per GPU, env shards, ONE all-reduce per outer ICRL iteration (RCCL over xGMI via torch.distributed
backend "nccl"; "gloo" on CPU for the tests).

The reference has no multi-device mode.  Rank r owns envs [r*N, (r+1)*N) (stream keys seed + global index),
runs its PPO epochs and its constraint-net update on its own shard, then everything that must stay common is averaged /
merged with a single flat float64 all-reduce:
    [ policy params | policy exp_avg | policy exp_avg_sq | cn params | cn exp_avg | cn exp_avg_sq | log_nu, m, v |
      obs_rms S | ret_rms S | cost_rms S ]        with S = (count, count*mean, count*(var + mean^2))
Parameters / moments are averaged (local-update data parallelism: the average of the ranks' parameter deltas is applied
once per outer iteration); running moments merge exactly: S_global = sum_g S_g - (G-1) * S_previous_common.
At <= 0.5 MB the message is latency-bound (7 x 153 GB/s xGMI links are irrelevant), hence ONE call.
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
    dist.init_process_group(backend)
    return rank, world


def reduce_device(default="cuda"):
    """where collectives' tensors live: device memory for RCCL, host memory for gloo."""
    return "cpu" if (dist.is_initialized() and dist.get_backend() == "gloo") else default


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


def pack(avg_tensors, rms_sums):
    flat = [t.detach().double().reshape(-1).cpu() if t.device.type != "cuda" else t.detach().double().reshape(-1) for t in avg_tensors]
    dev = flat[0].device if flat else torch.device("cpu")
    flat += [torch.as_tensor(s, dtype=torch.float64, device=dev) for s in rms_sums]
    return torch.cat(flat)


def allreduce_state(avg_tensors, rms_list, rms_prev_sums, world, group=None):
    """avg_tensors: tensors to average in place; rms_list: objects with .mean/.var/.count and .assign(mean, var, count);
    rms_prev_sums: their S at the last synchronisation.  ONE all-reduce.  Returns the new common S list."""
    sums = [moments_to_sums(r.mean, r.var, r.count) for r in rms_list]
    buf = pack(avg_tensors, sums)
    if world > 1:
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
    new_sums = []
    for r, s, prev in zip(rms_list, sums, rms_prev_sums):
        n = len(s)
        red = buf[off:off + n].cpu().numpy()
        off += n
        merged = merge_sums(red, prev, world) if world > 1 else red
        k = (n - 1) // 2
        mean, var, count = sums_to_moments(merged, k)
        r.assign(mean if k > 1 else mean[0], var if k > 1 else var[0], count)
        new_sums.append(merged)
    return new_sums
