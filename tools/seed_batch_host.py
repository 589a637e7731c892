"""where the HOST time of a lock-step seed-batch iteration goes: wall time inside the per-run host methods (no added synchronisation:
a method that waits for the GPU shows its wait)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from icrl_amd import seed_batch as SB, utils, ppo_lag, constraint_net, true_constraint_net, vec_env

S = int(os.environ.get("SEEDS", "32"))
sb = SB.SeedBatch([bench.config2(4, seed, 0, 1) for seed in range(S)])
sb.run(0, 2)
acc = {}
calls = {}
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj)}.{name}"
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
            calls[label] = calls.get(label, 0) + 1
    setattr(obj, name, w)
A = ppo_lag.PPOLagrangian
for n in ("_setup_learn", "_rollout_begin", "_rollout_end", "_train_begin", "_train_end", "train_readback", "_draw_permutations"):
    wrap(A, n, "agent." + n)
C = constraint_net.ConstraintNet
for n in ("_train_begin", "_train_end"):
    wrap(C, n, "cn." + n)
for n in ("compute_kl", "sample_result", "evaluate_result"):
    wrap(utils, n, "utils." + n)
wrap(SB, "mean_cost", "mean_cost")
wrap(SB, "sync_envs_normalization", "sync_envs_normalization")
_th = SB.SeedBatch.__dict__["_to_host"].__func__
def _timed_to_host(rows):
    t0 = time.perf_counter()
    try:
        return _th(rows)
    finally:
        acc["batch._to_host"] = acc.get("batch._to_host", 0.0) + time.perf_counter() - t0
SB.SeedBatch._to_host = staticmethod(_timed_to_host)
for n in ("_launch_rollouts", "_launch_trains", "_launch_episodes", "_launch_cn_trains", "_episodes", "_learn"):
    wrap(SB.SeedBatch, n, "batch." + n)
wrap(utils.EpisodeRun, "prepare", "EpisodeRun.prepare"); wrap(utils.EpisodeRun, "finish", "EpisodeRun.finish"); wrap(utils.EpisodeRun, "__init__", "EpisodeRun.__init__")
torch.cuda.synchronize(); t0 = time.perf_counter()
sb.run(2, 2)
tot = time.perf_counter() - t0
print(f"S={S}: {1e3 * tot / 2:.1f} ms per iteration")
print("  calls per iteration:", {k: c / 2 for k, c in calls.items() if k.startswith("batch._launch") or k.startswith("EpisodeRun")})
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:38s} {1e3 * v / 2:8.2f} ms")
