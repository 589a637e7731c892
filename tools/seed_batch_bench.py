"""aggregate env-steps/s of S independent ICRL runs (BASELINE configs[1] each) sharing one MI355X."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from icrl_amd import seed_batch as SB

for S in [int(x) for x in os.environ.get("SEEDS", "1,4,8,16,32").split(",")]:
    states = SB.setup_runs([bench.config2(4, seed, 0, 1) for seed in range(S)])
    SB.run_iterations(states, 0, 1)
    steps0 = sum(st["timesteps"] for st in states)
    _, dt = SB.run_iterations(states, 1, 2)
    steps = sum(st["timesteps"] for st in states) - steps0
    print(f"S={S:3d}: 2 iterations of every run in {dt:6.2f} s -> {steps / dt / 1e6:7.3f} M env-steps/s aggregate ({steps / dt / S / 1e3:7.1f} k per run)", flush=True)
    del states
    torch.cuda.empty_cache()
