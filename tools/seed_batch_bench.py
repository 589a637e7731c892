"""aggregate env-steps/s of S independent ICRL runs (BASELINE configs[1] each) sharing one MI355X inside the launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from icrl_amd import seed_batch as SB

for S in [int(x) for x in os.environ.get("SEEDS", "1,8,32,64").split(",")]:
    sb = SB.SeedBatch([bench.config2(4 + int(os.environ.get('ITERS', '2')), seed, 0, 1) for seed in range(S)])
    sb.run(0, 1)
    steps0 = sum(st["timesteps"] for st in sb.states)
    n_it = int(os.environ.get("ITERS", "2"))
    _, dt = sb.run(1, n_it)
    steps = sum(st["timesteps"] for st in sb.states) - steps0
    print(f"S={S:3d}: {n_it} iterations of every run in {dt:6.2f} s -> {steps / dt / 1e6:7.3f} M env-steps/s aggregate ({steps / dt / S / 1e3:7.1f} k per run)", flush=True)
    if os.environ.get("PHASES"):       # where one lock-step iteration goes (host clock, synchronised at the phase boundaries)
        import icrl_amd.seed_batch as M
        t = {}
        orig = {k: getattr(M.SeedBatch, k) for k in ("_learn", "_episodes", "_launch_cn_trains")}
        def timed(name, fn):
            def w(self, *a, **k):
                torch.cuda.synchronize(); t0 = time.time()
                r = fn(self, *a, **k)
                torch.cuda.synchronize(); t[name] = t.get(name, 0.0) + time.time() - t0
                return r
            return w
        for k, fn in orig.items():
            setattr(M.SeedBatch, k, timed(k, fn))
        t0 = time.time(); sb.run(3, 1); tot = time.time() - t0
        for k, fn in orig.items():
            setattr(M.SeedBatch, k, fn)
        print("      one iteration %.1f ms: " % (1e3 * tot) + ", ".join(f"{k} {1e3 * v:.1f} ms" for k, v in t.items()), flush=True)
    del sb
    torch.cuda.empty_cache()
