"""where a lock-step iteration of S runs goes: the batched rollout launch, the batched update launch and the host work around them,
each timed with a device synchronisation on both sides (tools only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from icrl_amd import seed_batch as SB, _lib
if os.environ.get("ICRL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ICRL_LIB"])

def sync():
    torch.cuda.synchronize(); return time.time()

for S in [int(x) for x in os.environ.get("SEEDS", "1,8,16,32,64").split(",")]:
    sb = SB.SeedBatch([bench.config2(4, seed, 0, 1) for seed in range(S)])
    sb.run(0, 1)
    sts = sb.states
    agents = [st["agent"] for st in sts]
    for a in agents:
        a._setup_learn(200000, True)
    res = {}
    for rep in range(2):
        t0 = sync()
        jobs = [a._rollout_begin(None, a.rollout_buffer, a.n_steps, None, zero_buffer=False) for a in agents]
        t1 = sync()
        sb._launch_rollouts(agents, jobs)
        t2 = sync()
        tj = []
        for a, j in zip(agents, jobs):
            a._rollout_end(j, a.env, None, a.rollout_buffer, a.n_steps)
            tj.append(a._train_begin(None))
        t3 = sync()
        sb._launch_trains(agents, tj)
        t4 = sync()
        host = sb._to_host([a.train_readback() for a in agents])
        for a, j, h in zip(agents, tj, host):
            a._train_end(j, host=h)
        t5 = sync()
        res = dict(rollout_begin=t1 - t0, rollout=t2 - t1, train_begin=t3 - t2, train=t4 - t3, train_end=t5 - t4)
    steps = int(host[0][1])
    if os.environ.get("PROF"):      # phase timers of the update (cycles per step; policy | reward-value workgroup of run 0 and of the last run)
        import numpy as np
        for a in agents:
            a.profile_phases = 1
        jobs = [a._rollout_begin(None, a.rollout_buffer, a.n_steps, None, zero_buffer=False) for a in agents]
        sb._launch_rollouts(agents, jobs)
        tj = []
        for a, j in zip(agents, jobs):
            a._rollout_end(j, a.env, None, a.rollout_buffer, a.n_steps)
            tj.append(a._train_begin(None))
        sb._launch_trains(agents, tj)
        host2 = sb._to_host([a.train_readback() for a in agents])
        for a, j, h in zip(agents, tj, host2):
            a._train_end(j, host=h)
        for who, a in (("run 0", agents[0]), ("last run", agents[-1])):
            st = a._train_ws["stats"].cpu().numpy()
            print(f"      {who}: cycles/step fwd|loss|bwd|wgrad|norm|wait|adam, policy {np.round(st[12:19])} reward-value {np.round(st[19:26])}", flush=True)
        for a in agents:
            a.profile_phases = 0
    print(f"S={S:3d}: " + ", ".join(f"{k} {1e3 * v:7.1f} ms" for k, v in res.items()) + f"; update {1e6 * res['train'] / steps:6.2f} us/step ({steps} steps), rollout {1e6 * res['rollout'] / 2048:6.2f} us/step", flush=True)
    if S == 1:      # the same run through the single-run entry points (argument blocks by value): is the batched form as fast?
        a = agents[0]
        for rep in range(2):
            t0 = sync()
            j = a._rollout_begin(None, a.rollout_buffer, a.n_steps, None, zero_buffer=False)
            a._rollout_launch(j)
            t1 = sync()
            a._rollout_end(j, a.env, None, a.rollout_buffer, a.n_steps)
            tj = a._train_begin(None)
            t2 = sync()
            a._train_launch(tj)
            t3 = sync()
            h = a.train_readback().cpu().numpy().reshape(-1)
            a._train_end(tj, host=h)
        print(f"      solo entry points: rollout {1e6 * (t1 - t0) / 2048:6.2f} us/step, update {1e6 * (t3 - t2) / int(h[1]):6.2f} us/step ({int(h[1])} steps)", flush=True)
    del sb, sts, agents
    torch.cuda.empty_cache()
