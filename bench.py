"""bench.py — env-steps/s of the ICRL outer loop (HCWithPos-v0 shapes, synthetic env) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W [--config {1,3,4}] [--mode {shards,seeds}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE outer ICRL iteration of BASELINE.json configs[1] (README.md:38 flags of the reference):
64 vectorised envs per GPU, forward_timesteps 2e5 (= 2 rollouts of 64 x 2048 + 2 PPO-Lagrangian updates of 10 epochs x
2048 minibatches of 64), nominal sampling (10 episodes), constraint-net update (10 iterations, 10 000 nominal + 5 000
expert rows), evaluation (10 episodes on the Test env) and both KL metrics.  Nothing is skipped inside the timed region.
Weak scaling: every rank owns its own 64 envs; one all-reduce of parameters / moments per outer iteration.
`value` includes the reference's target-KL early stops (ppo_lag.py:293-297): later outer iterations stop their epoch loops early, so
env-steps/s moves with the iteration count; `us_per_optimizer_step` (roofline_ppo) is the invariant, `optimizer_steps_per_iteration`
and `early_stop_fraction` say how much of the 2 x 20 480 steps an iteration executed.
  --config 1       (default at --gpus 1) BASELINE configs[1]: 64 envs per GPU
  --config 3       (default at --gpus > 1) BASELINE configs[3]'s per-GPU shard: HCWithPos, 2048 envs over 8 GPUs = 256 envs per GPU
  --config 4       BASELINE configs[4]: AntWall -> AntBroken constraint transfer (cpg, README.md:78 flags of the reference): frozen cost
                   net, 4096 envs over 8 GPUs = 512 envs per GPU; a "step" = one rollout (512 x 2048 env steps) + PPO-Lagrangian update of
                   learn(), value = env-steps/s of learn(); ranks all-reduce once per rollout + update
  --mode seeds     north_star's other fan-out: rank r runs an independent ICRL run with seed s + r, NO collective; value = sum over ranks
With one GPU the line also carries, as extras that are never `value`: `configs2` (BASELINE configs[2]: AntWall-v0, 256 envs, constraint
net [40, 40], batch 128, 20 epochs; README.md:50 flags), `configs3_one_gpu` (configs[3]'s 2048 envs whole on this GPU) and `configs4`
(configs[4]'s 512-env shard), each timed the same way.

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

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves
F32_MFMA_PEAK_TFLOPS = 157.3   # whole chip, 256 CUs


def _cfg(argv, rank, world):
    from icrl_amd.icrl import build_parser
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=rank, world_size=world, save_dir=None)
    return types.SimpleNamespace(**cfg)


def config2(n_iters_total, seed, rank, world, envs=64):
    """BASELINE configs[1] (README.md:38 flags of the reference); envs = 256: the per-GPU shard of configs[3]."""
    ex = os.path.join(ROOT, "tests/golden/expert_hc.npz")
    return _cfg(["icrl", "-er", "10", "-ep", ex, "-tk", "0.01", "-cl", "20", "-bi", "10",
                 "-ft", "2e5", "-ni", "30", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9",
                 "-crc", "0.5", "-psis", "-ctkno", "2.5", "-nt", str(envs), "-s", str(seed), "-v", "0", "--expert_agent_path", ex], rank, world)


def antwall_expert_path():
    """configs[2] uses 45 expert rollouts of 500 steps (22 500 x 121); the committed fixture holds 5.  Synthetic data of the full
    shape: the fixture's rollouts repeated 9 times with a deterministic perturbation, written once per process under /tmp."""
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"icrl_bench_expert_ant_{os.getpid()}.npz")
    if not os.path.exists(path):
        d = np.load(os.path.join(ROOT, "tests/golden/expert_ant.npz"))
        rng = np.random.RandomState(45)
        obs = np.concatenate([d["observations"] + 0.01 * rng.randn(*d["observations"].shape) for _ in range(9)])
        acs = np.concatenate([np.clip(d["actions"] + 0.01 * rng.randn(*d["actions"].shape).astype(np.float32), -1, 1) for _ in range(9)])
        extra = {k: d[k] for k in d.files if k.startswith("policy/")}
        np.savez(path, observations=obs, actions=acs.astype(np.float32), rewards=np.tile(d["rewards"], 9), lengths=np.tile(d["lengths"], 9), **extra)
    return path


def config_antwall(seed, rank, world, envs=256):
    """BASELINE configs[2] exactly as README.md:50 gives it: AntWall-v0, 256 envs, n_steps 2048, batch 128, 20 epochs, lr 3e-5,
    clip 0.4, lambdas 0.9, nu0 0.1, nu-lr 0.05, target_kl 0.02, constraint net [40, 40], 45 expert / nominal rollouts."""
    ex = antwall_expert_path()
    return _cfg(["icrl", "-ep", ex, "--expert_agent_path", ex, "-er", "45", "-cl", "40", "40", "-clr", "0.005", "-aclr", "0.9", "-crc", "0.6",
                 "-bi", "5", "-ft", "2e5", "-ni", "20", "-tei", "AntWall-v0", "-eei", "AntWallTest-v0", "--batch_size", "128",
                 "--reward_gae_lambda", "0.9", "--cost_gae_lambda", "0.9", "--n_epochs", "20", "--learning_rate", "3e-5", "--clip_range", "0.4",
                 "-piv", "0.1", "-plr", "0.05", "-psis", "-tk", "0.02", "-ctkno", "2.5", "-nt", str(envs), "-s", str(seed), "-v", "0"], rank, world)


def update_flops(O, A, H, B):
    """fp32 flops of ONE optimiser step, the ALGORITHMIC count: per network forward 2 B (O H + H H + H n),
    backward 2 B (O H [dW1] + 2 H H [dH1, dW2] + 2 H n [dH2, dWh]); n = A for the policy head, 1 for the critics; no d/dx of layer 1."""
    net = lambda n: 2 * B * (2 * O * H + 3 * H * H + 3 * H * n)
    return net(A) + 2 * net(1)


def timed_iterations(cfg, warmup, steps, world=1):
    """setup + `warmup` untimed + `steps` timed outer iterations of icrl_amd.icrl (identical calls to icrl()); returns the state,
    wall seconds and env steps of the timed part."""
    import torch.distributed as dist
    from icrl_amd import icrl as I
    st = I.setup(cfg)
    for it in range(warmup):
        I.outer_iteration(st, it)
    st["agent"].gae_events = []
    st["agent"].train_events = []
    st["sync_events"], st["sync_host_ms"] = [], []          # around the iteration's one collective (icrl_amd/icrl.py: synchronise)
    st["rollout_events"] = []
    st["agent"].rollout_events = st["rollout_events"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    steps0 = st["timesteps"]
    for it in range(warmup, warmup + steps):
        I.outer_iteration(st, it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    return st, time.time() - t0, st["timesteps"] - steps0


def update_summary(st, cfg):
    tr_ms = [e0.elapsed_time(e1) for e0, e1, _ in st["agent"].train_events]
    tr_steps = [n for _, _, n in st["agent"].train_events]
    n_mb = -(-cfg.n_steps * cfg.num_threads // cfg.batch_size)
    full = len(tr_steps) * cfg.n_epochs * n_mb
    us = 1e3 * float(np.sum(tr_ms)) / max(1, int(np.sum(tr_steps)))
    ro = [e0.elapsed_time(e1) for e0, e1 in st["agent"].rollout_events]
    return dict(us_per_optimizer_step=us, optimizer_steps=int(np.sum(tr_steps)), full_steps=full,
                us_per_rollout_step=(1e3 * float(np.mean(ro)) / cfg.n_steps) if ro else None)


def no_early_stop(env_steps, dt, u):
    """env-steps/s the same iterations would give with every epoch of every train() executed (target_kl never stopping a loop):
    the executed optimiser steps are replaced by the full count at the measured time per step, everything else as timed."""
    t_update = u["optimizer_steps"] * u["us_per_optimizer_step"] * 1e-6
    return env_steps / (dt - t_update + u["full_steps"] * u["us_per_optimizer_step"] * 1e-6)


def configs3_one_gpu_leg(seed, steps=1, warmup=1):
    """EXTRA: BASELINE configs[3]'s 2048 HCWithPos envs WHOLE on this GPU (rollout_multi_kernel, 10 epochs x 65 536 minibatches)."""
    cfg = config2(steps + warmup, seed, 0, 1, 2048)
    st, dt, env_steps = timed_iterations(cfg, warmup, steps)
    u = update_summary(st, cfg)
    out = dict(workload="HCWithPos-v0 ICRL, BASELINE configs[3]'s 2048 envs on ONE GPU: 1 rollout of 2048 x 2048 + PPO-Lag update of 10 epochs x "
                        "65 536 minibatches of 64 per outer iteration", value=round(env_steps / dt, 1), unit="env-steps/s", steps=steps, warmup=warmup,
               ms_per_step=round(1e3 * dt / steps, 2), us_per_optimizer_step=round(u["us_per_optimizer_step"], 2),
               us_per_rollout_step=None if u["us_per_rollout_step"] is None else round(u["us_per_rollout_step"], 2),
               optimizer_steps_per_iteration=round(u["optimizer_steps"] / steps, 1), early_stop_fraction=round(1.0 - u["optimizer_steps"] / max(1, u["full_steps"]), 4),
               value_no_early_stop=round(no_early_stop(env_steps, dt, u), 1))
    del st
    torch.cuda.empty_cache()
    return out


def generic_shape_leg(seed):
    """EXTRA: the generic-shape path (csrc/generic.hip) — what a user of `-sl / -pl / -rvl / -cvl` with anything but three two-layer
    branches of <= 64 units gets: HC shapes, 64 envs, `-sl 64 -pl 128 128 -rvl 64 -cvl 64 64 64`, batch 64; one rollout of 256 steps
    and one update of 2 epochs, timed with events on their stream.  Not a BASELINE config."""
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    N, T, B, E = 64, 256, 64, 2
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "hc", seed)))
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    arch = [64, dict(pi=[128, 128], vf=[64], cvf=[64, 64, 64])]
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=E, seed=seed, permutation="device", target_kl=None,
                          policy_kwargs=dict(net_arch=arch))
    agent._setup_learn(3 * N * T)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost"); agent.train()      # warm-up (module load)
    ev[0].record(); agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost"); ev[1].record()
    ev[2].record(); agent.train(); ev[3].record()
    torch.cuda.synchronize()
    steps = E * (N * T // B)
    return dict(workload=f"HC shapes, {N} envs, net_arch {arch}, batch {B}: policy_generic / gen_* kernels behind the same entry points",
                us_per_rollout_step=round(1e3 * ev[0].elapsed_time(ev[1]) / T, 1), us_per_optimizer_step=round(1e3 * ev[2].elapsed_time(ev[3]) / steps, 1),
                optimizer_steps=steps, rollout_steps=T)


def config_cpg(seed, rank, world, envs=512):
    """BASELINE configs[4] exactly as README.md:78 gives it (AntWallBroken-v0, frozen constraint net through ConstraintNet.load,
    batch 128, 20 epochs, lr 3e-5, clip 0.4, reward lambda 0.9, -plr 1.0, -tk 0.01); 4096 envs over 8 GPUs = 512 per GPU."""
    from icrl_amd.cpg import build_parser
    cfg = vars(build_parser().parse_args(["cpg", "--cn_path", os.path.join(ROOT, "tests/golden/cn_antbroken.npz"), "-tei", "AntWallBroken-v0",
                                          "-eei", "AntWallBrokenTest-v0", "-tk", "0.01", "--batch_size", "128", "--reward_gae_lambda", "0.9",
                                          "--n_epochs", "20", "--learning_rate", "3e-5", "--clip_range", "0.4", "-t", "2e6", "-plr", "1.0",
                                          "-nt", str(envs), "-s", str(seed), "-v", "0"]))
    cfg.update(rank=rank, world_size=world, save_dir=None)
    return types.SimpleNamespace(**cfg)


def timed_cpg(cfg, warmup, steps, world=1):
    """cpg (icrl_amd/cpg.py): setup, one untimed learn() of `warmup` rollouts + updates, one timed learn() of `steps` (the
    reference's cpg is a single learn() call, icrl/cpg.py:203; learn() resets the envs and its step count on every call)."""
    import torch.distributed as dist
    from icrl_amd import cpg as C
    model, cb, learn_cost, _ = C.setup(cfg, log=None)
    per = cfg.num_threads * cfg.n_steps
    if warmup > 0:
        model.learn(total_timesteps=warmup * per, cost_function=learn_cost, callback=cb)
    model.gae_events, model.train_events, model.rollout_events = [], [], []
    model.sync_events = None
    for c in cb.callbacks:
        if hasattr(c, "sync_events"):
            c.sync_events = model.sync_events = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    model.learn(total_timesteps=steps * per, cost_function=learn_cost, callback=cb)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    return model, time.time() - t0, float(model.num_timesteps)


# round 6: obs 65..128, minibatches of 65..128 rows (csrc/ppo_train_quarters2.hip; the row-owning kernel's two-workgroup form before: 18.0 us per step)
ANT_UPDATE_KERNEL = "ppo_train_quarters2_kernel (four workgroups per network, 32 rows each as two row tiles per wave: 12 CUs)"


def cpg_summary(model, cfg, dt, env_steps, steps, warmup):
    st = dict(agent=model)
    u = update_summary(st, cfg)
    fl = update_flops(113, 8, 64, cfg.batch_size)
    tf = fl / (u["us_per_optimizer_step"] * 1e-6) / 1e12
    return u, dict(us_per_optimizer_step=round(u["us_per_optimizer_step"], 2),
                   us_per_rollout_step=None if u["us_per_rollout_step"] is None else round(u["us_per_rollout_step"], 2),
                   optimizer_steps_per_iteration=round(u["optimizer_steps"] / steps, 1),
                   early_stop_fraction=round(1.0 - u["optimizer_steps"] / max(1, u["full_steps"]), 4),
                   value_no_early_stop=round(no_early_stop(env_steps, dt, u), 1),
                   update_kernel=ANT_UPDATE_KERNEL, rollout_kernel="rollout_wide_kernel",
                   update_tflops=round(tf, 4), update_frac_of_12cu_fp32_mfma_peak=round(tf / (F32_MFMA_PEAK_TFLOPS * 12 / 256), 4))


CPG_WORKLOAD = ("AntWall -> AntBroken constraint transfer (cpg, BASELINE configs[4], README.md:78 flags): AntWallBroken-v0, frozen constraint net "
                "(ConstraintNet.load of the AntBroken checkpoint), {n} envs per GPU, n_steps 2048, batch 128, 20 epochs, lr 3e-5, clip 0.4, "
                "-plr 1.0, target_kl 0.01 early stops INCLUDED; a step = one rollout + PPO-Lagrangian update of learn() with its "
                "callbacks (5 evaluation episodes per rollout)")


def configs4_leg(seed, steps=1, warmup=1):
    """EXTRA: BASELINE configs[4]'s per-GPU shard (512 envs) on this GPU."""
    cfg = config_cpg(seed, 0, 1)
    model, dt, env_steps = timed_cpg(cfg, warmup, steps)
    _, extra = cpg_summary(model, cfg, dt, env_steps, steps, warmup)
    out = dict(workload=CPG_WORKLOAD.format(n=cfg.num_threads), value=round(env_steps / dt, 1), unit="env-steps/s", steps=steps, warmup=warmup,
               ms_per_step=round(1e3 * dt / steps, 2), **extra)
    del model
    torch.cuda.empty_cache()
    return out


def configs2_leg(seed, steps=2, warmup=1):
    """EXTRA, never the headline value: BASELINE configs[2] on this GPU, `steps` timed outer iterations."""
    cfg = config_antwall(seed, 0, 1)
    st, dt, env_steps = timed_iterations(cfg, warmup, steps)
    u = update_summary(st, cfg)
    fl = update_flops(113, 8, 64, cfg.batch_size)
    tf = fl / (u["us_per_optimizer_step"] * 1e-6) / 1e12
    out = dict(workload="AntWall-v0 ICRL (BASELINE configs[2], README.md:50 flags): 256 envs, n_steps 2048, batch 128, 20 epochs, "
                        "constraint net [40, 40], 45 nominal + 10 eval episodes, 5 backward iterations on 22 500 + 22 500 rows of width 121",
               value=round(env_steps / dt, 1), unit="env-steps/s", steps=steps, warmup=warmup, ms_per_step=round(1e3 * dt / steps, 2),
               us_per_optimizer_step=round(u["us_per_optimizer_step"], 2), us_per_rollout_step=None if u["us_per_rollout_step"] is None else round(u["us_per_rollout_step"], 2),
               optimizer_steps_per_iteration=round(u["optimizer_steps"] / steps, 1), early_stop_fraction=round(1.0 - u["optimizer_steps"] / max(1, u["full_steps"]), 4),
               value_no_early_stop=round(no_early_stop(env_steps, dt, u), 1),
               update_kernel=ANT_UPDATE_KERNEL, rollout_kernel="rollout_wide_kernel",
               update_tflops=round(tf, 4), update_frac_of_12cu_fp32_mfma_peak=round(tf / (F32_MFMA_PEAK_TFLOPS * 12 / 256), 4))
    del st
    torch.cuda.empty_cache()
    return out


def allreduce_summary(events, per):
    """device time (pack -> RCCL all-reduce -> unpack, events on the launch stream) of the collective, per outer iteration / per rollout + update."""
    if not events:
        return None
    ms = [e0.elapsed_time(e1) for e0, e1 in events]
    return dict(ms_per_iteration=round(float(np.sum(ms)) / max(1, per), 3), calls=len(ms), ms_per_call=round(float(np.mean(ms)), 3))


def scale_anchor_leg(seed):
    """What the N > 1 lines must be divided by: `bench.py --gpus N>1` quotes BASELINE configs[3] (256 envs per GPU) or, with
    --config 4, configs[4] (512 per GPU); the N = 1 headline is configs[1] (64 envs).  Here ONE rank runs exactly the per-GPU workload of those
    lines through the MULTI-RANK code path: a 1-rank `nccl` (= RCCL) process group, allreduce_state(..., force_collective=True) once per outer
    iteration / per rollout + update — same config2(...) / config_cpg(...), same timing as the N > 1 line.  per_gpu_value(N) / anchor.value is the
    weak-scaling efficiency."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    out = {}
    try:
        cfg = config2(2, seed, 0, 1, 256)
        cfg.force_collective = True
        st, dt, env_steps = timed_iterations(cfg, 1, 1)
        u = update_summary(st, cfg)
        out["configs3"] = dict(workload="BASELINE configs[3]'s per-GPU shard: HCWithPos-v0 ICRL, 256 envs, one rank through the multi-rank path "
                                        "(1-rank RCCL group, one flat float64 all-reduce per outer iteration)", envs_per_gpu=256,
                               value=round(env_steps / dt, 1), unit="env-steps/s", steps=1, warmup=1, ms_per_step=round(1e3 * dt, 2),
                               us_per_optimizer_step=round(u["us_per_optimizer_step"], 2),
                               us_per_rollout_step=None if u["us_per_rollout_step"] is None else round(u["us_per_rollout_step"], 2),
                               early_stop_fraction=round(1.0 - u["optimizer_steps"] / max(1, u["full_steps"]), 4),
                               allreduce=allreduce_summary(st["sync_events"], 1), allreduce_host_ms=[round(x, 3) for x in st["sync_host_ms"]])
        del st
        torch.cuda.empty_cache()
        cfg = config_cpg(seed, 0, 1)
        cfg.force_collective = True
        model, dt, env_steps = timed_cpg(cfg, 1, 1)
        u, extra = cpg_summary(model, cfg, dt, env_steps, 1, 1)
        out["configs4"] = dict(workload="BASELINE configs[4]'s per-GPU shard: cpg on AntWallBroken-v0, 512 envs, one rank through the multi-rank path "
                                        "(1-rank RCCL group, one all-reduce per rollout + update)", envs_per_gpu=512,
                               value=round(env_steps / dt, 1), unit="env-steps/s", steps=1, warmup=1, ms_per_step=round(1e3 * dt, 2),
                               us_per_optimizer_step=extra["us_per_optimizer_step"], us_per_rollout_step=extra["us_per_rollout_step"],
                               early_stop_fraction=extra["early_stop_fraction"], allreduce=allreduce_summary(model.sync_events, 1))
        del model
        torch.cuda.empty_cache()
    finally:
        dist.destroy_process_group()
    out["how_to_read"] = ("`bench.py --gpus N` (N > 1) prints per_gpu_value = value / N for configs[3] (default) or configs[4] (--config 4): divide it by "
                          "scale_anchor.configs3.value / scale_anchor.configs4.value of THIS line, not by the headline `value` (configs[1], 64 envs)")
    return out


_GAE_BUFFERS = {}


def _gae_buffers(n_max):
    """inputs / outputs of the sweep, allocated once at the largest size (9 x 1.07 GB at 131 072 x 2048); smaller points use their heads."""
    if _GAE_BUFFERS.get("n", 0) < n_max:
        dev = torch.device("cuda")
        _GAE_BUFFERS.clear()
        torch.cuda.empty_cache()
        _GAE_BUFFERS.update(n=n_max, ins=[torch.randn(n_max, device=dev) for _ in range(4)] + [(torch.rand(n_max, device=dev) < 0.001).float()],
                            outs=[torch.empty(n_max, device=dev) for _ in range(4)])
    return _GAE_BUFFERS


def gae_sweep_point(N=131072, T=2048, reps=48):
    """icrl_gae_dual_ws (the entry point RolloutBufferWithCost calls, with the workspace the library asks for) at N envs x T rows.  At
    131 072 envs: working set 9.7 GB >> 256 MB Infinity Cache, the HBM-streaming regime (four columns per lane; bit-exact like the
    one-column scan); 8 192 .. 65 472 envs: the register-resident split scan (round 6).  In the same process, on the same stream and buffers,
    interleaved with the GAE launches: `icrl_debug_stream_ref` mode 0 (the streaming launch shape and the 5-loads-4-stores traffic of the GAE
    kernel without its recurrence: the same 36 B per transition) and mode 1 (a flat float4 copy, 32 B per transition) — what THIS box streams
    at this size, so that the kernel can be judged apart from the box's HBM rate."""
    from icrl_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda")
    b = _gae_buffers(T * N)
    ins = [x[:T * N].view(T, N) for x in b["ins"]]
    outs = [x[:T * N].view(T, N) for x in b["outs"]]
    lv = [torch.randn(N, device=dev) for _ in range(2)]
    ld = torch.zeros(N, dtype=torch.uint8, device=dev)
    ws = torch.zeros(int(L.icrl_gae_dual_ws_bytes(T, N)) // 8, dtype=torch.int64, device=dev)
    args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
    cargs = [_lib.ptr(x) for x in (*ins, *outs)]
    st = _lib.current_stream()
    launch = {"gae": lambda: L.icrl_gae_dual_ws(*args, T, N, 0.99, 0.95, 0.99, 0.95, 0, _lib.ptr(ws), ws.numel() * 8, st),
              "copy": lambda: L.icrl_debug_stream_ref(*cargs, T, N, 1, st)}
    if N % 4 == 0 and N >= 65536:       # (the streaming shape's grid is N / 256 one-wave workgroups: no reference below a full chip)
        launch["ref"] = lambda: L.icrl_debug_stream_ref(*cargs, T, N, 0, st)
    for f in launch.values():
        for _ in range(3):
            _lib.check(f(), "gae sweep")
    torch.cuda.synchronize()
    assert int(ws.view(torch.int32)[-1].item()) == 0
    ms = {k: [] for k in launch}
    rounds, per = 6, max(1, reps // 6)
    for _ in range(rounds):                     # interleaved rounds: a drifting HBM clock hits all three alike
        for k, f in launch.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms[k].append(e0.elapsed_time(e1) / per)
    t = {k: float(np.median(v)) for k, v in ms.items()}      # (median over the rounds: one disturbed round of one kernel would move a mean by 5-20 %)
    gbs = lambda nbytes, k: nbytes / (t[k] * 1e-3) / 1e9
    tiles = (N + 63) // 64
    kernel = ("gae_dual_x4_kernel<4,1>" if tiles >= 2048 and N % 4 == 0 else "gae_dual_kernel<1,16,nt,4>" if tiles >= 1024 else
              f"gae_dual_regsplit_kernel ({tiles} x {-(-T // 128)} workgroups)" if T <= 2048 else "gae_dual_kernel")
    return dict(envs=N, T=T, bytes=T * N * 36, us=t["gae"] * 1e3, achieved=gbs(T * N * 36, "gae"), kernel=kernel,
                ref_gbs=gbs(T * N * 36, "ref") if "ref" in t else None, ref_us=t["ref"] * 1e3 if "ref" in t else None,
                copy_gbs=gbs(T * N * 32, "copy"), copy_us=t["copy"] * 1e3,
                launches=rounds * per, spread={k: [round(min(v) * 1e3), round(max(v) * 1e3)] for k, v in ms.items()})


def gae_size_sweep(sizes=(64, 512, 4096, 8192, 32768, 65536), T=2048):
    """SURVEY 8(d)'s size sweep with the current kernels: every point through the default heuristic of icrl_gae_dual_ws."""
    rows = []
    for N in sizes:
        p = gae_sweep_point(N, T, reps=48 if N >= 4096 else 96)
        rows.append(dict(envs=N, T=T, kernel=p["kernel"], us=round(p["us"], 1), achieved=round(p["achieved"], 1), frac=round(p["achieved"] / HBM_PEAK_GBS, 4),
                         frac_of_copy=None if p["ref_gbs"] is None else round(p["achieved"] / p["ref_gbs"], 4),
                         frac_of_flat_copy=round(p["achieved"] / p["copy_gbs"], 4), bytes=p["bytes"],
                         regime="cache-resident" if p["bytes"] * 20 // 36 <= 256e6 else "hbm"))
    return rows


def gae_roofline(sweep, in_loop, size_sweep=None):
    """the `roofline` object: the dual-GAE kernel at the HBM-streaming launch shape, timed live with events on the launch stream;
    `traffic` from the committed PMC passes of the same launch shape (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes)."""
    import glob
    traffic = None
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*gae_pmc.json")))      # latest round's passes
    pmc_path = pmc_files[-1] if pmc_files else ""
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        # the figure is a committed measurement of gae.hip's streaming kernel: it may ride along only while that source is the one measured
        import hashlib
        gae_sha = hashlib.sha256(open(os.path.join(ROOT, "icrl_amd", "csrc", "gae.hip"), "rb").read()).hexdigest()[:16]
        if pmc["T"] == sweep["T"] and pmc["N"] == sweep["envs"] and pmc.get("gae_hip_sha16") == gae_sha:
            traffic = int(pmc["traffic_bytes"])
    r = dict(kernel="gae_dual_x4_kernel", bound="hbm", achieved=round(sweep["achieved"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
             frac=round(sweep["achieved"] / HBM_PEAK_GBS, 4), traffic=traffic,
             traffic_source=(f"not measured in this run: PMC passes of the same launch shape committed as profiles/{os.path.basename(pmc_path)} "
                             "(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes)") if traffic is not None else None,
             at=f"T={sweep['T']}, N={sweep['envs']} envs: {sweep['bytes'] / 1e9:.2f} GB algorithmic (36 B/transition), "
                f"{sweep['us']:.0f} us/launch — far beyond the 256 MB Infinity Cache; four columns per lane (16-byte accesses), "
                f"{sweep['envs'] // 256} one-wave workgroups")
    # self-normalisation: the same bytes through the same launch shape without the recurrence, and a flat float4 copy, timed
    # in this process on this stream between the GAE launches (HBM rate differs box to box by more than the kernel does)
    r.update(copy_gbs=round(sweep["ref_gbs"], 1), frac_of_copy=round(sweep["achieved"] / sweep["ref_gbs"], 4),
             copy_kernel="stream_ref_x4_kernel<4> (icrl_debug_stream_ref mode 0): grid, 5 non-temporal 16-byte loads + 4 stores per lane and row and "
                         f"bytes of the GAE launch, no recurrence: {sweep['ref_us']:.0f} us/launch",
             flat_copy_gbs=round(sweep["copy_gbs"], 1), frac_of_flat_copy=round(sweep["achieved"] / sweep["copy_gbs"], 4),
             flat_copy_kernel=f"stream_copy_kernel (mode 1): four grid-stride float4 copies, {sweep['T'] * sweep['envs'] * 32 / 1e9:.2f} GB, {sweep['copy_us']:.0f} us",
             launches_timed=sweep["launches"], us_min_max_per_round=sweep["spread"])
    if in_loop is not None:
        r["in_loop"] = in_loop
    if size_sweep is not None:
        r["sweep"] = size_sweep + [dict(envs=sweep["envs"], T=sweep["T"], kernel=sweep["kernel"], us=round(sweep["us"], 1), achieved=round(sweep["achieved"], 1),
                                        frac=round(sweep["achieved"] / HBM_PEAK_GBS, 4), frac_of_copy=round(sweep["achieved"] / sweep["ref_gbs"], 4),
                                        frac_of_flat_copy=round(sweep["achieved"] / sweep["copy_gbs"], 4), bytes=sweep["bytes"], regime="hbm")]
    return r


def seed_batch_leg(sizes=(8, 32, 64)):
    """EXTRA, not the headline value: S independent runs of the same workload (seeds 100 ..) sharing the GPU INSIDE the launches
    (icrl_amd/seed_batch.py: one host thread, every phase one launch with grid.y = run; every run bit-identical to its solo run,
    tests/test_seed_batch_gpu.py).  One warm-up iteration, then 2 timed iterations of every run."""
    from icrl_amd import seed_batch as SB
    res = {}
    for S in sizes:
        sb = SB.SeedBatch([config2(4, 100 + s_, 0, 1) for s_ in range(S)])
        sb.run(0, 1)                                          # warm-up
        steps0 = sum(st["timesteps"] for st in sb.states)
        _, dt = sb.run(1, 2)
        res[str(S)] = round((sum(st["timesteps"] for st in sb.states) - steps0) / dt, 1)
        del sb
        torch.cuda.empty_cache()
    return dict(aggregate_env_steps_per_s=res, unit="env-steps/s", note="S independent ICRL runs (seeds) of the same workload on one "
                "MI355X in lock-step from one host thread: rollouts of all runs = one launch of 8 S persistent workgroups (8 envs each, the MLPs as fp32 MFMA tiles), updates = one launch of "
                "3 S persistent workgroups, 2 timed outer iterations each; not the headline value")


def cpu_baseline():
    """oracle CPU port (`oracle.loop.icrl_port`, pinned bit-for-bit to the reference's own icrl() by tests/golden/g8) timed on the
    host: ONE whole outer ICRL iteration of configs[1] at FULL size — 2 rollouts of 64 x 2048 + 2 PPO-Lagrangian updates of
    10 epochs x 2048 minibatches, 10 nominal + 10 evaluation episodes, constraint-net update, both KL metrics — on 1 torch thread
    (the faster setting for 64-wide MLPs; measured, no extrapolation; ~45 s).  The 8-thread figure the reference's own measurement
    used is taken on a bounded sample (n_steps 256, same work per env step and per minibatch) and reported beside it."""
    from oracle import loop as o_loop
    ex = np.load(os.path.join(ROOT, "tests/golden/expert_hc.npz"))
    esd = {k[len("policy/"):]: ex[k] for k in ex.files if k.startswith("policy/")}

    def run(T, threads):
        cfg = dict(train_env_id="HCWithPos-v0", eval_env_id="HCWithPosTest-v0", num_threads=64, seed=0, n_steps=T, batch_size=64, n_epochs=10,
                   target_kl=0.01, cn_layers=(20,), cn_learning_rate=0.05, anneal_clr_by_factor=0.9, cn_reg_coeff=0.5,
                   per_step_importance_sampling=True, cn_target_kl_new_old=2.5, backward_iters=10, forward_timesteps=2 * 64 * T - 1,
                   n_iters=30, expert_rollouts=10)
        torch.set_num_threads(threads)
        m, steps, dt, _ = o_loop.icrl_port(cfg, ex["observations"][:5000], ex["actions"][:5000], esd, n_iters=1)
        return steps, m[0]["time/forward_s"], m[0]["time/rest_s"]

    steps, fwd, rest = run(2048, 1)
    value = steps / (fwd + rest)
    kk = min(8, os.cpu_count() or 1)
    s8, f8, r8 = run(256, kk)
    v8 = 8 * s8 / (8 * f8 + r8)
    ratio = 1382.0 / 944.0      # port vs reference, learn() of this shape in the build container
    return dict(value=value, unit="env-steps/s", cores=1, kind="port", value_1_thread=value,
                reference_equivalent=value / ratio,
                reference_equivalent_note="value / 1.46: in the build container the port's learn() of this shape runs 1 382 env-steps/s against the "
                                          "reference's 944 (BASELINE.md section 4); the reference itself cannot travel to the GPU box",
                **{f"value_{kk}_threads_sample": v8},
                sample=f"one whole outer ICRL iteration of the same workload at full size (n_steps 2048: {steps} env steps, 40 960 optimiser "
                       f"steps before target-KL stops) on 1 torch thread: forward step {fwd:.1f} s + sampling, constraint-net update, "
                       f"evaluation and KL metrics {rest:.1f} s, measured; the {kk}-thread figure from an n_steps 256 sample "
                       f"({s8} env steps: {f8:.1f} s + {r8:.1f} s, forward scaled by 8); host {os.cpu_count()} logical cores")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--config", type=int, default=None, choices=(1, 3, 4),
                    help="BASELINE configs[] index of the headline workload: 1 = 64 HCWithPos envs per GPU (default on one GPU), 3 = 256 per GPU = "
                         "2048 over 8 GPUs (default with --gpus > 1), 4 = the AntWall -> AntBroken transfer, 512 envs per GPU")
    ap.add_argument("--mode", default="shards", choices=("shards", "seeds"), help="shards: env shards + one all-reduce per outer iteration; seeds: independent runs, no collective")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_seed_batch", action="store_true")
    ap.add_argument("--no_configs2", action="store_true")
    ap.add_argument("--no_configs3", action="store_true")
    ap.add_argument("--no_configs4", action="store_true")
    ap.add_argument("--no_scale_anchor", action="store_true", help="skip the one-rank anchors of the N > 1 lines (configs[3] / configs[4] shards through a 1-rank RCCL group)")
    ap.add_argument("--no_generic", action="store_true", help="skip the generic-shape extra (a non-default net_arch at toy size)")
    ap.add_argument("--envs_per_gpu", type=int, default=None, help="override the env count per GPU (e.g. 2048: BASELINE configs[3] whole on one GPU)")
    a = ap.parse_args()

    # ONE JSON line on stdout, whatever the libraries print: RCCL writes its version banner to the C-level stdout when its first
    # communicator is created (5 lines, flushed at exit).  File descriptor 1 is pointed at stderr for the whole run; the line goes to a
    # duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        json_out.write(json.dumps(obj) + "\n")
        json_out.flush()

    import torch.distributed as dist
    from icrl_amd import distributed as D
    rank, world = D.init_from_env()
    if world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    # the multi-GPU default is a BASELINE config too: configs[3] = 2048 envs over 8 GPUs = 256 per GPU (64 per GPU x 8 would be none)
    config_id = a.config if a.config is not None else (1 if world == 1 else 3)

    def reduce_over_ranks(dt, env_steps):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=D.reduce_device())
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            s_ = torch.tensor([env_steps], dtype=torch.float64, device=D.reduce_device())
            dist.all_reduce(s_, op=dist.ReduceOp.SUM)
            return float(t.item()), float(s_.item())
        return dt, env_steps

    if config_id == 4:
        # ---- BASELINE configs[4]: one learn() of `steps` rollouts + updates per rank, all-reduce once per rollout + update
        seeds = a.mode == "seeds"
        cfg = config_cpg(a.seed + (rank if seeds else 0), 0 if seeds else rank, 1 if seeds else world, a.envs_per_gpu or 512)
        model, dt, env_steps = timed_cpg(cfg, a.warmup, a.steps, world)
        dt, env_steps = reduce_over_ranks(dt, env_steps)
        if rank != 0:
            return
        u, extra = cpg_summary(model, cfg, dt, env_steps, a.steps, a.warmup)
        sweep = gae_sweep_point()
        par = (f"env-shards x{world}, 1 all-reduce / rollout + update" if not seeds else f"independent seeds x{world}, no collective")
        out = dict(metric="env-steps/sec (cpg learn(), AntWallBroken-v0)", value=round(env_steps / dt, 1), unit="env-steps/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * dt / a.steps, 2), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=CPG_WORKLOAD.format(n=cfg.num_threads), envs_per_gpu=cfg.num_threads, n_steps=cfg.n_steps,
                               batch_size=cfg.batch_size, n_epochs=cfg.n_epochs, baseline_config=4, mode=a.mode, parallelism=par),
                   roofline=gae_roofline(sweep, None), **extra)
        out["cpu_baseline"] = None
        out["per_gpu_value"] = round(env_steps / dt / world, 1)
        out["scale_anchor_ref"] = "scale_anchor.configs4.value of the N = 1 line (`python bench.py`): the same 512-env shard on one rank through this code path"
        out["allreduce"] = allreduce_summary(getattr(model, "sync_events", None), a.steps)
        emit(out)
        return

    envs = a.envs_per_gpu or (64 if config_id == 1 else 256)
    if a.mode == "seeds":      # independent runs: every rank is a 1-rank job with its own seed; the only collectives are the timing ones below
        cfg = config2(a.steps + a.warmup, a.seed + rank, 0, 1, envs)
    else:
        cfg = config2(a.steps + a.warmup, a.seed, rank, world, envs)

    # ---- the outer loop, one iteration at a time (identical calls to icrl_amd.icrl.icrl; see that function)
    st, dt, env_steps = timed_iterations(cfg, a.warmup, a.steps, world)
    dt, env_steps = reduce_over_ranks(dt, env_steps)
    if os.environ.get("ICRL_BENCH_RANK_DUMP"):      # tests/test_bench_launch_gpu.py: which env keys a rank stepped, what it holds after the last reduce
        import hashlib
        ag, cn_, senv = st["agent"], st["constraint_net"], st["train_env"].unwrapped
        h = hashlib.sha256()
        for t_ in (ag.policy.params, ag.policy.exp_avg, ag.policy.exp_avg_sq, cn_.params, st["train_env"].obs_rms.d_mean, st["train_env"].obs_rms.d_var):
            h.update(t_.detach().cpu().numpy().tobytes())
        with open(os.path.join(os.environ["ICRL_BENCH_RANK_DUMP"], f"rank{rank}.json"), "w") as f:
            json.dump(dict(rank=rank, world=world, env_keys=(lambda k: [int(k.min()), int(k.max()) + 1, int(len(set(k.tolist())))])(senv.key.cpu().numpy().view(np.uint32)),
                           state_sha=h.hexdigest(), nu=float(ag.dual.nu().item()), adam_step=int(ag.policy.adam_step), env_steps=float(st["timesteps"])), f)
    if rank != 0:
        return

    # ---- roofline of the GAE kernel from the in-loop launches (events recorded on the launch stream)
    gae_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in st["agent"].gae_events]
    T, N = cfg.n_steps, cfg.num_threads
    gae_bytes = T * N * 36
    gae_ach = gae_bytes / (np.mean(gae_us) * 1e-6) / 1e9
    sweep = gae_sweep_point()
    size_sweep = gae_size_sweep() if world == 1 else None
    _GAE_BUFFERS.clear()
    torch.cuda.empty_cache()
    roofline = gae_roofline(sweep, dict(achieved=round(gae_ach, 1), frac=round(gae_ach / HBM_PEAK_GBS, 5), us_per_launch=round(float(np.mean(gae_us)), 1),
                                        launches=len(gae_us), bytes_per_launch=gae_bytes,
                                        note="the launch the loop itself makes (cache-resident, latency-bound): two-level scan, "
                                             "time axis split over workgroups x 8 waves, rows register-resident (gae_dual_regsplit_kernel)"), size_sweep)
    # ---- the PPO kernel: algorithmic flops per optimiser step (update_flops) / the events around icrl_ppo_lag_train
    u = update_summary(st, cfg)
    B = cfg.batch_size
    flops_step = update_flops(18, 6, 64, B)
    us_per_step = u["us_per_optimizer_step"]
    ppo_tflops = flops_step / (us_per_step * 1e-6) / 1e12
    quarters = os.environ.get("ICRL_QUARTERS", "1") != "0"      # (csrc/ppo_train.hip: quarters_default)
    n_cu = 12 if quarters else 6      # ppo_train_halves_kernel: four workgroups per network (round 6; two in round 5, the wave-pair kernel of rounds 2-4 ran on 3)
    roofline_ppo = dict(kernel=f"ppo_train_halves_kernel<2, false, 18, {4 if quarters else 2}>", bound="mfma", achieved=round(ppo_tflops, 4), peak=round(F32_MFMA_PEAK_TFLOPS * n_cu / 256, 3),
                        unit="TFLOP/s", frac=round(ppo_tflops / (F32_MFMA_PEAK_TFLOPS * n_cu / 256), 4), compute_units=n_cu,
                        chip_peak=F32_MFMA_PEAK_TFLOPS, frac_chip=round(ppo_tflops / F32_MFMA_PEAK_TFLOPS, 5),
                        us_per_optimizer_step=round(us_per_step, 2), optimizer_steps=u["optimizer_steps"], flops_per_step=flops_step,
                        us_per_rollout_step=None if u["us_per_rollout_step"] is None else round(u["us_per_rollout_step"], 2),
                        note="dependent optimiser steps of the reference algorithm, step LATENCY is what counts: 12 workgroups (four per MLP, 16 rows = one row tile of the "
                             "minibatch each, the four partial gradients exchanged through the XCD's L2 and summed in a fixed order by all four) = 12 of 256 CUs; peak = fp32 MFMA "
                             "rate of those 12 CUs — the same algorithmic flops on twice the compute units of round 5 in 0.96 x the time: the fraction of peak halves again, the "
                             "step shortens (6.74 -> 6.46 us); flops_per_step is the algorithmic count (DESIGN.md section 5), the padded 16x16x4 MFMA work actually issued is "
                             "35 instructions per wave and step (SQ counters: profiles/r06_train_pmc.md)")
    par = (f"env-shards x{world}, 1 all-reduce / outer iteration" if a.mode == "shards" else f"independent seeds x{world}, no collective")
    out = dict(metric="env-steps/sec (ICRL outer loop, HCWithPos-v0)", value=round(env_steps / dt, 1), unit="env-steps/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * dt / a.steps, 2), higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=f"HCWithPos-v0 ICRL (BASELINE configs[{config_id}]" + ("" if config_id == 1 else ": 2048 envs sharded 8 ways = 256 per GPU") +
                                    f"): {N} vectorised envs per GPU, n_steps 2048, forward_timesteps 2e5 ({-(-cfg.forward_timesteps // (N * T))} rollout(s) + "
                                    f"PPO-Lag update(s) of 10 epochs x {N * T // B} minibatches of 64, target_kl 0.01 early stops INCLUDED: value moves with the "
                                    "iteration count, us_per_optimizer_step does not), 10 nominal + 10 eval episodes, constraint net [20] x 10 backward iterations",
                           envs_per_gpu=N, n_steps=T, batch_size=B, n_epochs=cfg.n_epochs, forward_timesteps=cfg.forward_timesteps,
                           baseline_config=config_id, mode=a.mode, parallelism=par),
               optimizer_steps_per_iteration=round(u["optimizer_steps"] / a.steps, 1),
               early_stop_fraction=round(1.0 - u["optimizer_steps"] / max(1, u["full_steps"]), 4),
               value_no_early_stop=round(no_early_stop(env_steps / world, dt, u) * world, 1),      # (rank 0's step counts stand for every rank's)
               roofline=roofline, roofline_ppo=roofline_ppo)
    out["cpu_baseline"] = None if (a.no_cpu_baseline or world > 1) else cpu_baseline()      # reported at N = 1 only
    out["per_gpu_value"] = round(env_steps / dt / world, 1)
    if world > 1:
        out["scale_anchor_ref"] = ("scale_anchor.configs3.value of the N = 1 line (`python bench.py`): the same 256-env shard on one rank through this code path"
                                   if config_id == 3 and envs == 256 else "a one-rank run of this line's own flags")
        out["allreduce"] = allreduce_summary(st["sync_events"], a.steps)
    if out["cpu_baseline"] is not None:       # no published number exists for this metric: the ratio to the reference-equivalent CPU figure
        out["vs_baseline"] = round(out["value"] / out["cpu_baseline"]["reference_equivalent"], 1)
        out["vs_baseline_note"] = "value / cpu_baseline.reference_equivalent (the reference's CPU path on this host, via the port: BASELINE.md section 4); BASELINE.md publishes no number for this metric"
    del st
    torch.cuda.empty_cache()
    if world == 1 and not a.no_configs2:
        out["configs2"] = configs2_leg(a.seed)
    if world == 1 and not a.no_configs3 and config_id == 1 and a.envs_per_gpu is None:
        out["configs3_one_gpu"] = configs3_one_gpu_leg(a.seed)
    if world == 1 and not a.no_configs4:
        out["configs4"] = configs4_leg(a.seed)
    if world == 1 and not a.no_generic:
        out["generic_shape"] = generic_shape_leg(a.seed)
    if world == 1 and not a.no_scale_anchor and config_id == 1 and a.envs_per_gpu is None:
        try:
            out["scale_anchor"] = scale_anchor_leg(a.seed)
        except Exception as e:      # (a box without a working RCCL must not cost the headline line)
            out["scale_anchor"] = dict(error=f"{type(e).__name__}: {e}")
    if world == 1 and not a.no_seed_batch:
        out["seed_batch"] = seed_batch_leg()
    emit(out)


if __name__ == "__main__":
    main()
