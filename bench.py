"""bench.py — env-steps/s of the ICRL outer loop (HCWithPos-v0 shapes, synthetic env) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE outer ICRL iteration of BASELINE.json configs[1] (README.md:38 flags of the reference):
64 vectorised envs per GPU, forward_timesteps 2e5 (= 2 rollouts of 64 x 2048 + 2 PPO-Lagrangian updates of 10 epochs x
2048 minibatches of 64), nominal sampling (10 episodes), constraint-net update (10 iterations, 10 000 nominal + 5 000
expert rows), evaluation (10 episodes on the Test env) and both KL metrics.  Nothing is skipped inside the timed region.
Weak scaling: every rank owns its own 64 envs; one all-reduce of parameters / moments per outer iteration.

The JSON line also carries
  roofline      the dual-GAE kernel (the kernel BASELINE.json's metric names): algorithmic 36 B/transition / live event timing
                of the in-loop launches, plus the same kernel at N = 131 072 envs where the 9.7 GB working set streams from HBM
  roofline_ppo  the persistent PPO-Lagrangian kernel (where the time goes): fp32 MFMA flops vs the 3 CUs it occupies
  cpu_baseline  the oracle CPU port (same algorithmic structure as the reference) timed on a bounded sample on the host.
"""
import argparse
import json
import os
import sys
import time
import types

# the seed-batch leg runs up to 32 independent ICRL runs on 32 HIP streams: ROCm multiplexes streams onto 4 hardware queues unless
# told otherwise (measured: aggregate throughput saturates at 4 runs); must be set before the runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves
F32_MFMA_PEAK_TFLOPS = 157.3   # whole chip, 256 CUs


def config2(n_iters_total, seed, rank, world):
    from icrl_amd.icrl import build_parser
    argv = ["icrl", "-er", "10", "-ep", os.path.join(ROOT, "tests/golden/expert_hc.npz"), "-tk", "0.01", "-cl", "20", "-bi", "10",
            "-ft", "2e5", "-ni", "30", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9",
            "-crc", "0.5", "-psis", "-ctkno", "2.5", "-nt", "64", "-s", str(seed), "-v", "0",
            "--expert_agent_path", os.path.join(ROOT, "tests/golden/expert_hc.npz")]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=rank, world_size=world, save_dir=None)
    return types.SimpleNamespace(**cfg)


def gae_sweep_point(N=131072, T=2048, reps=20):
    """the same entry point at 131 072 envs: working set 9.7 GB >> 256 MB Infinity Cache, the HBM-streaming regime (the library
    picks the four-columns-per-lane streaming shape from 65 536 envs on; bit-exact like the one-column scan)."""
    from icrl_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda")
    ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.001).float()]
    lv = [torch.randn(N, device=dev) for _ in range(2)]
    ld = torch.zeros(N, dtype=torch.uint8, device=dev)
    outs = [torch.empty(T, N, device=dev) for _ in range(4)]
    args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
    st = _lib.current_stream()
    for _ in range(5):
        L.icrl_gae_dual(*args, T, N, 0.99, 0.95, 0.99, 0.95, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.icrl_gae_dual(*args, T, N, 0.99, 0.95, 0.99, 0.95, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(envs=N, T=T, bytes=T * N * 36, us=ms * 1e3, achieved=T * N * 36 / (ms * 1e-3) / 1e9)


def seed_batch_leg(sizes=(8, 32)):
    """EXTRA, not the headline value: S independent runs of the same workload (seeds 100 ..) sharing the GPU, each on its own
    HIP stream / host thread (icrl_amd/seed_batch.py; every run bit-identical to its solo run, tests/test_seed_batch_gpu.py).
    One warm-up iteration, then 2 timed iterations of every run."""
    from icrl_amd import seed_batch as SB
    res = {}
    for S in sizes:
        states = SB.setup_runs([config2(4, 100 + s_, 0, 1) for s_ in range(S)])
        SB.run_iterations(states, 0, 1)                       # warm-up
        steps0 = sum(st["timesteps"] for st in states)
        _, dt = SB.run_iterations(states, 1, 2)
        res[str(S)] = round((sum(st["timesteps"] for st in states) - steps0) / dt, 1)
        del states
        torch.cuda.empty_cache()
    return dict(aggregate_env_steps_per_s=res, unit="env-steps/s", note="S independent ICRL runs (seeds) of the same workload on one "
                "MI355X, one HIP stream + host thread per run, 2 timed outer iterations each; not the headline value")


def cpu_baseline():
    """oracle CPU port (`oracle.loop.icrl_port`, pinned bit-for-bit to the reference's own icrl() by tests/golden/g8) timed on
    a BOUNDED sample of the same workload: one whole outer ICRL iteration of configs[1] — forward step (rollouts + PPO-Lagrangian
    updates), 10 nominal + 10 evaluation episodes, constraint-net update, both KL metrics — with n_steps 256 instead of 2048.
    The forward step's cost is proportional to n_steps (same work per env step and per minibatch), the rest does not depend on
    it, so the full-size iteration time is 8 x t_forward + t_rest; value = 2 x 64 x 2048 env steps / that.  Run with the 8 torch
    intra-op threads the reference's own measurement used, and with 1."""
    from oracle import loop as o_loop
    ex = np.load(os.path.join(ROOT, "tests/golden/expert_hc.npz"))
    esd = {k[len("policy/"):]: ex[k] for k in ex.files if k.startswith("policy/")}
    T, scale = 256, 8
    cfg = dict(train_env_id="HCWithPos-v0", eval_env_id="HCWithPosTest-v0", num_threads=64, seed=0, n_steps=T, batch_size=64, n_epochs=10,
               target_kl=0.01, cn_layers=(20,), cn_learning_rate=0.05, anneal_clr_by_factor=0.9, cn_reg_coeff=0.5,
               per_step_importance_sampling=True, cn_target_kl_new_old=2.5, backward_iters=10, forward_timesteps=2 * 64 * T - 1,
               n_iters=30, expert_rollouts=10)
    res = {}
    for threads in (min(8, os.cpu_count() or 1), 1):
        torch.set_num_threads(threads)
        m, steps, dt, _ = o_loop.icrl_port(cfg, ex["observations"][:5000], ex["actions"][:5000], esd, n_iters=1)
        full = scale * m[0]["time/forward_s"] + m[0]["time/rest_s"]
        res[threads] = (scale * steps / full, m[0]["time/forward_s"], m[0]["time/rest_s"], steps)
    k = max(res, key=lambda n_: res[n_][0])               # the faster of the two thread counts is the baseline (fair to the CPU)
    kk = max(res)
    return dict(value=res[k][0], unit="env-steps/s", cores=k, kind="port", value_1_thread=res[1][0], **{f"value_{kk}_threads": res[kk][0]},
                sample=f"one whole outer ICRL iteration of the same workload with n_steps={T} instead of 2048 ({res[k][3]} env steps of "
                       f"the forward step: {res[kk][1]:.1f} s on {kk} torch threads / {res[1][1]:.1f} s on 1; sampling + constraint-net update "
                       f"+ evaluation + KL metrics: {res[kk][2]:.1f} s / {res[1][2]:.1f} s), extrapolated to n_steps 2048 as 8 x forward + rest; "
                       f"host {os.cpu_count()} logical cores.  In the build container the port's learn() step runs 1.46x the "
                       f"reference's (BASELINE.md section 2: 1382 vs 944 env-steps/s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_seed_batch", action="store_true")
    a = ap.parse_args()

    import torch.distributed as dist
    from icrl_amd import distributed as D
    rank, world = D.init_from_env()
    if world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())

    from icrl_amd import icrl as I, logger, utils
    from icrl_amd.vec_env import sync_envs_normalization
    cfg = config2(a.steps + a.warmup, a.seed, rank, world)

    # ---- the outer loop, one iteration at a time (identical calls to icrl_amd.icrl.icrl; see that function)
    st = I.setup(cfg)
    for it in range(a.warmup):
        I.outer_iteration(st, it)
    st["agent"].gae_events = []
    st["agent"].train_events = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    steps0 = st["timesteps"]
    for it in range(a.warmup, a.warmup + a.steps):
        I.outer_iteration(st, it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    env_steps = st["timesteps"] - steps0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=D.reduce_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([env_steps], dtype=torch.float64, device=D.reduce_device())
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        env_steps = float(s.item())
    if rank != 0:
        return

    # ---- roofline of the GAE kernel from the in-loop launches (events recorded on the launch stream)
    gae_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in st["agent"].gae_events]
    T, N = cfg.n_steps, cfg.num_threads
    gae_bytes = T * N * 36
    gae_ach = gae_bytes / (np.mean(gae_us) * 1e-6) / 1e9
    sweep = gae_sweep_point()
    traffic = None   # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), same shape
    import glob
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*gae_pmc.json")))      # latest round's passes
    pmc_path = pmc_files[-1] if pmc_files else ""
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        if pmc["T"] == sweep["T"] and pmc["N"] == sweep["envs"]:
            traffic = int(pmc["traffic_bytes"])
    roofline = dict(kernel="gae_dual_x4_kernel", bound="hbm", achieved=round(sweep["achieved"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(sweep["achieved"] / HBM_PEAK_GBS, 4), traffic=traffic,
                    traffic_source=(f"not measured in this run: PMC passes of the same launch shape committed as profiles/{os.path.basename(pmc_path)} "
                                    "(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes)") if traffic is not None else None,
                    at=f"T={sweep['T']}, N={sweep['envs']} envs: {sweep['bytes'] / 1e9:.2f} GB algorithmic (36 B/transition), "
                       f"{sweep['us']:.0f} us/launch — far beyond the 256 MB Infinity Cache; four columns per lane (16-byte accesses), "
                       f"{sweep['envs'] // 256} one-wave workgroups",
                    in_loop=dict(achieved=round(gae_ach, 1), frac=round(gae_ach / HBM_PEAK_GBS, 5), us_per_launch=round(float(np.mean(gae_us)), 1),
                                 launches=len(gae_us), bytes_per_launch=gae_bytes,
                                 note="the launch the loop itself makes (4.7 MB, cache-resident, latency-bound): two-level scan, "
                                      "time axis split over 16 workgroups x 8 waves (gae_dual_split_kernel)"))
    # ---- the PPO kernel: flops of the 8 GEMMs per optimiser step x 3 nets, from the events around icrl_ppo_lag_train
    tr_ms = [e0.elapsed_time(e1) for e0, e1, _ in st["agent"].train_events]
    tr_steps = [n for _, _, n in st["agent"].train_events]
    O, A, H, B = 18, 6, 64, cfg.batch_size
    flops_step = 3 * 2 * B * (O * H + H * H) * 3 + 2 * B * H * (A + 2) * 3      # fwd + 2x bwd of the three MLPs (+ heads)
    us_per_step = 1e3 * float(np.sum(tr_ms)) / max(1, int(np.sum(tr_steps)))
    ppo_tflops = flops_step / (us_per_step * 1e-6) / 1e12
    roofline_ppo = dict(kernel="ppo_train_pairs_kernel", bound="mfma", achieved=round(ppo_tflops, 4), peak=round(F32_MFMA_PEAK_TFLOPS * 3 / 256, 3),
                        unit="TFLOP/s", frac=round(ppo_tflops / (F32_MFMA_PEAK_TFLOPS * 3 / 256), 4),
                        chip_peak=F32_MFMA_PEAK_TFLOPS, frac_chip=round(ppo_tflops / F32_MFMA_PEAK_TFLOPS, 5),
                        us_per_optimizer_step=round(us_per_step, 2), optimizer_steps=int(np.sum(tr_steps)),
                        note="dependent optimiser steps of the reference algorithm: 3 workgroups (one per MLP) = 3 of 256 CUs; "
                             "peak = fp32 MFMA rate of those 3 CUs; the padded 16x16x4 MFMA work actually issued is 2 x 148 instructions "
                             "per SIMD and step = 9.5k of the ~22k cycles of a step (SQ counters: profiles/r02_train_pmc.md)")
    out = dict(metric="env-steps/sec (ICRL outer loop, HCWithPos-v0)", value=round(env_steps / dt, 1), unit="env-steps/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * dt / a.steps, 2), higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload="HCWithPos-v0 ICRL (BASELINE configs[1]): 64 vectorised envs per GPU, n_steps 2048, "
                                    "forward_timesteps 2e5 (2 rollouts + 2 PPO-Lag updates of 10 epochs x 2048 minibatches of 64), "
                                    "10 nominal + 10 eval episodes, constraint net [20] x 10 backward iterations, target_kl 0.01",
                           envs_per_gpu=N, n_steps=T, batch_size=B, n_epochs=cfg.n_epochs, forward_timesteps=cfg.forward_timesteps,
                           parallelism=f"env-shards x{world}, 1 all-reduce / outer iteration"),
               roofline=roofline, roofline_ppo=roofline_ppo)
    out["cpu_baseline"] = None if (a.no_cpu_baseline or world > 1) else cpu_baseline()      # reported at N = 1 only
    if world == 1 and not a.no_seed_batch:
        out["seed_batch"] = seed_batch_leg()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
