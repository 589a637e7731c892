"""Where one outer ICRL iteration of the benchmark configuration spends its wall time (run on the GPU box):
each section of icrl_amd.icrl.outer_iteration is wrapped with a synchronise + host timer (SYNC=1, default: section = host + GPU
time, serialised) or only a host timer (SYNC=0: what the host thread itself spends, with the GPU running behind it)."""
import os, sys, time, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from icrl_amd import icrl as I, utils, ppo_lag, constraint_net

SYNC = os.environ.get("SYNC", "1") == "1"
acc = collections.OrderedDict()


def wrap(owner, name, label):
    fn = getattr(owner, name)

    def timed(*a, **k):
        if SYNC:
            torch.cuda.synchronize()
        t0 = time.time()
        out = fn(*a, **k)
        if SYNC:
            torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + time.time() - t0
        return out
    setattr(owner, name, timed)


wrap(ppo_lag.PPOLagrangian, "collect_rollouts", "rollout (launch + GAE)")
wrap(ppo_lag.PPOLagrangian, "_draw_permutations", "  of which permutations (host)")
wrap(ppo_lag.PPOLagrangian, "train", "update")
wrap(utils, "sample_from_agent", "nominal episodes")
wrap(constraint_net.ConstraintNet, "train", "constraint-net update")
wrap(utils, "evaluate_policy", "evaluation episodes")
wrap(utils, "compute_kl", "KL metrics")
wrap(I, "mean_cost", "true cost")
wrap(I, "synchronise", "synchronise")

cfg = bench.config2(4, 0, 0, 1)
st = I.setup(cfg)
I.outer_iteration(st, 0)
acc.clear()
torch.cuda.synchronize(); t0 = time.time()
n = 3
for it in range(1, 1 + n):
    I.outer_iteration(st, it)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"SYNC={int(SYNC)}  {1e3 * dt / n:.1f} ms per outer iteration")
tot = 0.0
for k, v in acc.items():
    print(f"  {k:34s} {1e3 * v / n:8.2f} ms")
    if not k.startswith("  "):
        tot += v
print(f"  {'(outside the sections)':34s} {1e3 * (dt - tot) / n:8.2f} ms")
