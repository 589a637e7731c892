"""CPU only: the whole-run record of BASELINE configs[1] (VERDICT r5 missing #2) — tests/golden/g19_whole_run_hc.npz.

north_star's result criterion is "returns, constraint-violation rate, Lagrange multiplier trajectory on identical seeds".  Over MANY outer
iterations two correct fp32 executions of the reference algorithm separate (tools/calibrate_drift.py: ~1e-3 x lr x steps per train()),
so a single pair of runs says little; this tool records what the CPU port (oracle.loop.icrl_port, pinned to the reference's own icrl()
by g8) logs per outer iteration at FULL size — HCWithPos-v0, 64 envs x 2048 steps, README.md:38 flags, SeededStreams(19) — once
undisturbed and several times with a rounding-size disturbance (every initial parameter moved by -1 / 0 / +1 float32 ulp, the rows of every
minibatch reversed / rotated: another summation order in every optimiser step), and stores per metric and iteration the undisturbed value and
the port-vs-port minimum / maximum.  tests/test_icrl_trajectory_gpu.py::test_icrl_hc_whole_run_vs_port_band runs the HIP loop on the same
streams and initial weights and must stay inside that band (widened by its own width) at every iteration.

    python tools/gen_whole_run.py run <variant> <out.json> [n_iters]      variant: base | ulp | ulpm | rnd<k> | rev | rot<k>
    python tools/gen_whole_run.py band base.json other.json ...          -> tests/golden/g19_whole_run_hc.npz + a markdown table on stdout

Only arrays are stored (metrics, initial weights, the flag list); ref: icrl/icrl.py:199-304.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

STREAM_SEED = 19
N_ENVS, N_STEPS, BATCH = 64, 2048, 64
# the reference's README.md:38 flags for HCWithPos ICRL (what bench.py: config2 passes), as oracle.loop.icrl_port's keys
CFG = dict(train_env_id="HCWithPos-v0", eval_env_id="HCWithPosTest-v0", num_threads=N_ENVS, seed=0, n_steps=N_STEPS, batch_size=BATCH, n_epochs=10,
           target_kl=0.01, cn_layers=(20,), cn_learning_rate=0.05, anneal_clr_by_factor=0.9, cn_reg_coeff=0.5, per_step_importance_sampling=True,
           cn_target_kl_new_old=2.5, backward_iters=10, forward_timesteps=200000, n_iters=30, expert_rollouts=10)
ARGV = ["icrl", "-er", "10", "-tk", "0.01", "-cl", "20", "-bi", "10", "-ft", "2e5", "-ni", "30", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0",
        "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis", "-ctkno", "2.5", "-nt", str(N_ENVS), "-s", "0", "-v", "0"]


class PermutedRows:
    """SeededStreams whose minibatches keep their rows but change their order (rev / rot<k>)."""

    def __init__(self, inner, variant):
        self.inner, self.variant = inner, variant

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def permutation(self, epoch, n):
        p = np.asarray(self.inner.permutation(epoch, n))
        m = (n // BATCH) * BATCH
        head = p[:m].reshape(-1, BATCH)
        head = head[:, ::-1] if self.variant == "rev" else np.roll(head, int(self.variant[3:]), axis=1)
        return np.concatenate([head.reshape(-1), p[m:]])


def base_init():
    """the port's own initial networks for CFG (seed 0): built exactly as icrl_port builds them, no iteration run."""
    from oracle import loop as o_loop
    ex = np.load(os.path.join(ROOT, "tests/golden/expert_hc.npz"))
    _, _, _, objs = o_loop.icrl_port(CFG, ex["observations"], ex["actions"], None, n_iters=0)
    return (dict((k, v.detach().numpy().copy()) for k, v in objs["agent"].policy.params.items()),
            dict((k, v.detach().numpy().copy()) for k, v in objs["cn"].params.items()))


def disturb(sd, variant):
    out = {}
    rng = np.random.RandomState(int(variant[3:])) if variant.startswith("rnd") else None
    for k, v in sd.items():
        t = torch.as_tensor(v)
        up, dn = torch.nextafter(t, torch.full_like(t, float("inf"))), torch.nextafter(t, torch.full_like(t, float("-inf")))
        if variant == "ulp":
            t = up
        elif variant == "ulpm":
            t = dn
        elif rng is not None:
            m = torch.as_tensor(rng.randint(-1, 2, size=tuple(t.shape)))
            t = torch.where(m > 0, up, torch.where(m < 0, dn, t))
        out[k] = t.numpy().copy()
    return out


def run(variant, out_path, n_iters=10):
    from oracle import loop as o_loop
    from oracle.streams import SeededStreams
    torch.set_num_threads(1)
    ex = np.load(os.path.join(ROOT, "tests/golden/expert_hc.npz"))
    esd = {k[len("policy/"):]: ex[k] for k in ex.files if k.startswith("policy/")}
    w0, cn0 = base_init()
    streams = SeededStreams(STREAM_SEED)
    if variant in ("ulp", "ulpm") or variant.startswith("rnd"):
        init = dict(policy=disturb(w0, variant), cn=cn0)
    else:
        init = dict(policy=w0, cn=cn0)
        if variant != "base":
            streams = PermutedRows(streams, variant)
    t0 = time.time()
    rows = []

    def log(m):
        rows.append({k: float(v) for k, v in m.items() if np.ndim(v) == 0})
        print(variant, "iteration", int(m["iteration"]), "nu", round(m["forward/nu"], 6), "true/reward", round(m["true/reward"], 2), round(time.time() - t0, 1), "s", flush=True)
        json.dump(dict(variant=variant, metrics=rows), open(out_path, "w"))
    o_loop.icrl_port(CFG, ex["observations"], ex["actions"], esd, n_iters=n_iters, streams=streams, init=init, log=log)


SKIP = ("time/", "time(m)")


def band(base_path, others):
    base = json.load(open(base_path))["metrics"]
    runs = [json.load(open(p))["metrics"] for p in others]
    n_it = min([len(base)] + [len(r) for r in runs])
    keys = sorted(k for k in base[0] if not k.startswith(SKIP))
    val = np.array([[base[i][k] for k in keys] for i in range(n_it)])
    allv = np.array([[[r[i][k] for k in keys] for i in range(n_it)] for r in [base] + runs])
    lo, hi = np.nanmin(allv, axis=0), np.nanmax(allv, axis=0)
    w0, cn0 = base_init()
    out = os.path.join(ROOT, "tests/golden/g19_whole_run_hc.npz")
    np.savez_compressed(out, meta=np.array(repr(dict(torch=torch.__version__, numpy=np.__version__, runs=1 + len(runs), variants=[os.path.basename(p) for p in others]))),
                        argv=np.array(ARGV), stream_seed=STREAM_SEED, metric_keys=np.array(keys), base=val, lo=lo, hi=hi,
                        **{f"w0/{k}": v for k, v in w0.items()}, **{f"cn0/{k}": v for k, v in cn0.items()})
    print(f"wrote {out} ({os.path.getsize(out) / 1024:.0f} KB): {n_it} outer iterations, {len(keys)} metrics, {1 + len(runs)} runs of the CPU port\n")
    show = ("forward/nu", "forward/average_cost", "true/cost", "true/reward", "forward/early_stop_epoch", "forward/reward_explained_variance",
            "backward/cn_loss", "backward/kl_new_old", "backward/kl_old_new", "true/forward_kl", "true/reverse_kl")
    print("| iteration | " + " | ".join(show) + " |")
    print("|---|" + "---|" * len(show))
    for i in range(n_it):
        cells = []
        for k in show:
            j = keys.index(k)
            cells.append(f"{val[i, j]:.6g} [{lo[i, j]:.6g}, {hi[i, j]:.6g}]")
        print(f"| {i} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 10)
    else:
        band(sys.argv[2], sys.argv[3:])
