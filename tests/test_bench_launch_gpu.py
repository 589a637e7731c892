"""GPU: bench.py through the exact launch line the driver uses for N > 1 — `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` — with two ranks on ONE GPU
(ICRL_DIST_BACKEND=gloo: the reduction goes through host memory; RCCL itself needs >= 2 GPUs) at toy env counts: argument parsing,
the default multi-GPU config (BASELINE configs[3]), the timing barriers, max-over-ranks / sum-over-ranks, rank 0's one JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, nproc=2, timeout=900, dump_dir=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ICRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if dump_dir is not None:
        env["ICRL_BENCH_RANK_DUMP"] = str(dump_dir)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "1", "--warmup", "1"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines          # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


def _steps(r):
    """env steps of the timed region, from the (rounded) value and time of the line"""
    return r["value"] * r["ms_per_step"] * 1e-3 * r["steps"]


def test_default_multi_gpu_config_is_baseline_configs3():
    r = _launch(["--envs_per_gpu", "8"])
    assert r["n_gpus"] == 2 and r["steps"] == 1 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["baseline_config"] == 3 and r["config"]["mode"] == "shards" and "all-reduce" in r["config"]["parallelism"]
    # whole-job value: both ranks' env steps (forward_timesteps 2e5 -> 13 rollouts of 8 x 2048 per rank) / the slower rank's time
    assert r["value"] > 0 and _steps(r) == pytest.approx(2 * 13 * 8 * 2048, rel=1e-3)
    # (two processes share ONE GPU here, the second one exiting while rank 0 runs its sweep: the HBM rate is noise in this set-up —
    # 2.2-3.6 TB/s measured — and only its presence is checked; the rate is bench.py's business on a GPU of its own)
    assert r["roofline"]["bound"] == "hbm" and 0.05 < r["roofline"]["frac"] < 1.0 and r["cpu_baseline"] is None
    # the kernel against what THIS box, process and moment stream for the same bytes (far more robust to the shared GPU than `frac` — the
    # other rank's process exits while rank 0 sweeps: 0.68-1.0 measured in this set-up —, and it still catches a kernel at half its rate)
    assert r["roofline"]["copy_gbs"] > 0 and r["roofline"]["frac_of_copy"] > 0.5
    assert r["per_gpu_value"] == pytest.approx(r["value"] / 2, rel=1e-3) and r["allreduce"]["calls"] == 1
    assert "configs2" not in r and "seed_batch" not in r                  # extras are an N = 1 matter


def test_without_overrides_the_shard_is_256_envs():
    """no --envs_per_gpu: the per-GPU shard of configs[3] (2048 envs / 8 GPUs); checked on the parsed workload, one rank is enough."""
    r = _launch([], nproc=2)
    assert r["config"]["envs_per_gpu"] == 256 and r["config"]["baseline_config"] == 3
    assert _steps(r) == pytest.approx(2 * 256 * 2048, rel=1e-3)


def test_independent_seeds_mode():
    r = _launch(["--mode", "seeds", "--config", "1", "--envs_per_gpu", "8"])
    assert r["config"]["mode"] == "seeds" and "no collective" in r["config"]["parallelism"] and r["n_gpus"] == 2 and r["value"] > 0


def test_configs4_cpg_transfer_two_ranks():
    r = _launch(["--config", "4", "--envs_per_gpu", "32"])
    assert r["config"]["baseline_config"] == 4 and r["n_gpus"] == 2 and "cpg" in r["metric"]
    assert _steps(r) == pytest.approx(2 * 32 * 2048, rel=1e-3)      # one rollout + update per rank in the timed learn()
    assert r["us_per_optimizer_step"] > 0 and r["optimizer_steps_per_iteration"] > 0 and "all-reduce / rollout" in r["config"]["parallelism"]


def test_eight_ranks_own_eight_shards_and_agree_after_the_reduce(tmp_path):
    """The driver's 8-GPU launch line with eight ranks on ONE GPU (gloo transport, 8 envs per rank, forward_timesteps = one rollout): every rank
    steps its own env-key range [seed + 8 r, seed + 8 r + 8), the whole-job value counts all eight shards, the N > 1 line carries per_gpu_value /
    scale_anchor_ref / the measured all-reduce, and after the last outer iteration's single all-reduce all ranks hold the SAME policy, moments,
    constraint net, observation statistics and Lagrange multiplier."""
    r = _launch(["--envs_per_gpu", "8"], nproc=8, timeout=1500, dump_dir=tmp_path)
    assert r["n_gpus"] == 8 and r["config"]["baseline_config"] == 3 and r["scaling"] == "weak"
    assert r["per_gpu_value"] == pytest.approx(r["value"] / 8, rel=1e-3) and isinstance(r["scale_anchor_ref"], str)
    assert r["allreduce"]["calls"] == 1 and r["allreduce"]["ms_per_iteration"] > 0
    dumps = [json.load(open(os.path.join(tmp_path, f"rank{k}.json"))) for k in range(8)]
    ranges = sorted(tuple(d["env_keys"]) for d in dumps)
    assert ranges == [(8 * k, 8 * k + 8, 8) for k in range(8)], ranges          # seed 0: eight disjoint, contiguous key ranges of 8 distinct keys
    assert len({d["state_sha"] for d in dumps}) == 1 and len({d["nu"] for d in dumps}) == 1 and len({d["adam_step"] for d in dumps}) == 1
    assert _steps(r) == pytest.approx(sum(d["env_steps"] for d in dumps) / 2, rel=1e-3)      # (warm-up + timed iteration in every rank's count)
