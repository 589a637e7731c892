"""CPU only: the whole-run record of BASELINE configs[1] (VERDICT r5 missing #2) — tests/golden/g19_whole_run_hc.npz.

north_star's result criterion is "returns, constraint-violation rate, Lagrange multiplier trajectory on identical seeds".  Over MANY outer
iterations two correct fp32 executions of the reference algorithm separate (tools/calibrate_drift.py: ~1e-3 x lr x steps per train()),
so a single pair of runs says little; this tool records what the CPU port (oracle.loop.icrl_port, pinned to the reference's own icrl()
by g8) logs per outer iteration at FULL size — HCWithPos-v0, 64 envs x 2048 steps, README.md:38 flags, SeededStreams(19) — once
undisturbed and several times with a rounding-size disturbance (every initial parameter moved by -1 / 0 / +1 float32 ulp, the rows of every
minibatch reversed / rotated: another summation order in every optimiser step), and stores per metric and iteration the undisturbed value and
the port-vs-port minimum / maximum.  tests/test_icrl_trajectory_gpu.py::test_icrl_whole_run_vs_port_band runs the HIP loop on the same
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

# ---- the other two single-GPU BASELINE configs (round 6, second half): the flag list is the definition, the port's keys are read off it through
# the product's own parser (host code).  `run2 <config> <variant> out.json [n_iters]` / `band2 <config> base.json other.json ...`
CONFIGS = {
    # BASELINE configs[0]: README.md:25 — LapGridWorld, 1 env, n_steps 2000, 25 rollouts + updates per outer iteration
    "lgw": dict(golden="g20_whole_run_lgw", stream_seed=5, batch=64, uniform=True, expert="tests/golden/expert_lgw.npz", n_iters=6,
                argv=["icrl", "-er", "20", "-tei", "LGW-v0", "-eei", "CLGW-v0", "-tk", "0.01", "-cl", "20", "-clr", "0.003", "-ft", "0.5e5", "-ni", "10", "-bi", "20",
                      "-dno", "-dnr", "-dnc", "--n_steps", "2000", "-nt", "1", "-s", "0", "-v", "0"]),
    # BASELINE configs[2]: README.md:50 — AntWall, 256 envs, batch 128, 20 epochs, constraint net [40, 40], 45 expert / nominal rollouts (bench.py: config_antwall)
    "ant": dict(golden="g21_whole_run_ant", stream_seed=23, batch=128, expert="antwall45", n_iters=6,
                argv=["icrl", "-er", "45", "-cl", "40", "40", "-clr", "0.005", "-aclr", "0.9", "-crc", "0.6", "-bi", "5", "-ft", "2e5", "-ni", "20", "-tei", "AntWall-v0",
                      "-eei", "AntWallTest-v0", "--batch_size", "128", "--reward_gae_lambda", "0.9", "--cost_gae_lambda", "0.9", "--n_epochs", "20", "--learning_rate", "3e-5",
                      "--clip_range", "0.4", "-piv", "0.1", "-plr", "0.05", "-psis", "-tk", "0.02", "-ctkno", "2.5", "-nt", "256", "-s", "0", "-v", "0"]),
}


def antwall45_expert(path):
    """45 expert rollouts of 500 steps (22 500 x 121) from the committed 5: the fixture's rollouts nine times with a deterministic perturbation
    (what bench.py: antwall_expert_path writes) — both the tool and the GPU test build the file with this function."""
    if not os.path.exists(path):
        d = np.load(os.path.join(ROOT, "tests/golden/expert_ant.npz"))
        rng = np.random.RandomState(45)
        obs = np.concatenate([d["observations"] + 0.01 * rng.randn(*d["observations"].shape) for _ in range(9)])
        acs = np.concatenate([np.clip(d["actions"] + 0.01 * rng.randn(*d["actions"].shape).astype(np.float32), -1, 1) for _ in range(9)])
        extra = {k: d[k] for k in d.files if k.startswith("policy/")}
        np.savez(path, observations=obs, actions=acs.astype(np.float32), rewards=np.tile(d["rewards"], 9), lengths=np.tile(d["lengths"], 9), **extra)
    return path


def config2(name, tmp_dir="/tmp"):
    """(port cfg dict, expert obs, expert acs, expert policy state dict, expert path) of CONFIGS[name]."""
    from icrl_amd.icrl import build_parser          # host code only: the reference's flag names and defaults
    from icrl_amd import utils
    from oracle import loop as o_loop
    c = CONFIGS[name]
    ex_path = antwall45_expert(os.path.join(tmp_dir, "icrl_whole_run_expert_ant45.npz")) if c["expert"] == "antwall45" else os.path.join(ROOT, c["expert"])
    cfg = vars(build_parser().parse_args(c["argv"] + ["-ep", ex_path, "--expert_agent_path", ex_path]))
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    (eo, ea), _ = utils.load_expert_data(ex_path, cfg["expert_rollouts"])
    d = np.load(ex_path)
    esd = {k[len("policy/"):]: d[k] for k in d.files if k.startswith("policy/")}
    return port_cfg, eo, ea, esd, ex_path


class PermutedRows:
    """SeededStreams whose minibatches keep their rows but change their order (rev / rot<k>)."""

    def __init__(self, inner, variant, batch=BATCH):
        self.inner, self.variant, self.batch = inner, variant, batch

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def permutation(self, epoch, n):
        p = np.asarray(self.inner.permutation(epoch, n))
        m = (n // self.batch) * self.batch
        head = p[:m].reshape(-1, self.batch)
        head = head[:, ::-1] if self.variant == "rev" else np.roll(head, int(self.variant[3:]), axis=1)
        return np.concatenate([head.reshape(-1), p[m:]])


def base_init(name=None):
    """the port's own initial networks for CFG / CONFIGS[name] (seed 0): built exactly as icrl_port builds them, no iteration run."""
    from oracle import loop as o_loop
    if name is not None:
        port_cfg, eo, ea, _, _ = config2(name)
        _, _, _, objs = o_loop.icrl_port(port_cfg, eo, ea, None, n_iters=0)
        return (dict((k, v.detach().numpy().copy()) for k, v in objs["agent"].policy.params.items()),
                dict((k, v.detach().numpy().copy()) for k, v in objs["cn"].params.items()))
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


def run2(name, variant, out_path, n_iters=None):
    from oracle import loop as o_loop
    from oracle.streams import SeededStreams
    torch.set_num_threads(1)
    c = CONFIGS[name]
    port_cfg, eo, ea, esd, _ = config2(name)
    w0, cn0 = base_init(name)
    if variant.startswith("tanhe6"):     # a 1e-6-relative pseudo-random error in EVERY tanh evaluation (tools/calibrate_drift.py: the documented size of the
        orig, ph = torch.tanh, float(variant[6:] or 0)      # kernels' v_exp-based tanh against libm's) — the disturbance class that is representative of the HIP path where
        torch.tanh = lambda x: orig(x) * (1 + 1e-6 * torch.sin(12345.678 * x + ph))      # single discrete events (an action / a clip decision flipping) carry the drift
    streams = SeededStreams(c["stream_seed"], uniform=c.get("uniform", False))
    if variant in ("ulp", "ulpm") or variant.startswith("rnd"):
        init = dict(policy=disturb(w0, variant), cn=cn0)
    else:
        init = dict(policy=w0, cn=cn0)
        if variant not in ("base",) and not variant.startswith("tanhe6"):
            streams = PermutedRows(streams, variant, c["batch"])
    t0, rows = time.time(), []

    def log(m):
        rows.append({k: float(v) for k, v in m.items() if np.ndim(v) == 0})
        print(name, variant, "iteration", int(m["iteration"]), "nu", round(m["forward/nu"], 6), "true/reward", round(m["true/reward"], 2), "true/cost", round(m["true/cost"], 4),
              round(time.time() - t0, 1), "s", flush=True)
        json.dump(dict(variant=variant, config=name, metrics=rows), open(out_path, "w"))
    o_loop.icrl_port(port_cfg, eo, ea, esd, n_iters=n_iters or c["n_iters"], streams=streams, init=init, log=log)


# ---- BASELINE configs[4]'s per-GPU shard: cpg's learn() (icrl/cpg.py:203: ONE learn() call) on AntWallBroken-v0, 512 envs x 2048 steps, the reference's frozen
# AntBroken constraint net, README.md:78 flags — per ROLLOUT + update: what PPOLagrangian.train() logs.  `run3 <variant> out.json [n_rollouts]`, `band3 base.json ...`
CPG = dict(golden="g22_whole_run_cpg", stream_seed=31, N=512, T=2048, seed=0, n_rollouts=4, cn="tests/golden/cn_antbroken.npz",
           kw=dict(batch_size=128, n_epochs=20, target_kl=0.01, learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, penalty_learning_rate=1.0))


def cpg_port(w0=None):
    """(PortAgent, its EnvStack) of CPG; w0: initial policy weights (None: the port's own for the seed)."""
    from oracle import loop as o_loop, nets as o_nets
    z = np.load(os.path.join(ROOT, CPG["cn"]))
    ocn = o_nets.CostNet(113, 8, [int(h) for h in z["hidden_sizes"]], False, None, None, None, None, None)      # the off-by-one load(): no clipping, no normalisation
    ocn.load_state_dict({k[len("cn_network/"):]: z[k] for k in z.files if k.startswith("cn_network/")})
    stack = o_loop.make_stack(CPG["N"], "ant", CPG["seed"], broken=True); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=CPG["T"], seed=CPG["seed"], **CPG["kw"])
    if w0 is not None:
        port.policy.load_state_dict(w0)
    return port, stack


def run3(variant, out_path, n_rollouts=None):
    from oracle.streams import SeededStreams
    torch.set_num_threads(1)
    if variant.startswith("tanhe6"):
        orig, ph = torch.tanh, float(variant[6:] or 0)
        torch.tanh = lambda x: orig(x) * (1 + 1e-6 * torch.sin(12345.678 * x + ph))
    port, stack = cpg_port()
    w0 = {k: v.detach().numpy().copy() for k, v in port.policy.params.items()}
    if variant in ("ulp", "ulpm") or variant.startswith("rnd"):
        port.policy.load_state_dict(disturb(w0, variant))
    streams = SeededStreams(CPG["stream_seed"])
    if variant in ("rev",) or variant.startswith("rot"):
        streams = PermutedRows(streams, variant, CPG["kw"]["batch_size"])
    N, T = CPG["N"], CPG["T"]
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    rows, t0 = [], time.time()
    for k in range(n_rollouts or CPG["n_rollouts"]):
        port.collect_rollouts(streams.rollout_noise(T, N, port.act_dim))
        out = port.train(lambda e: streams.permutation(e, T * N))
        streams.consumed(min(int(out["train/early_stop_epoch"]) + 1, CPG["kw"]["n_epochs"]))
        rows.append({k_: float(v) for k_, v in out.items() if np.ndim(v) == 0 and k_.startswith("train/")})
        print("cpg", variant, "rollout", k, "nu", round(out["train/nu"], 6), "average_cost", round(out["train/average_cost"], 5), "epochs", out["train/early_stop_epoch"], round(time.time() - t0, 1), "s", flush=True)
        json.dump(dict(variant=variant, config="cpg", metrics=rows), open(out_path, "w"))


def band3(base_path, others):
    base = json.load(open(base_path))["metrics"]
    runs = [json.load(open(p))["metrics"] for p in others]
    n_it = min([len(base)] + [len(r) for r in runs])
    keys = sorted(base[0])
    val = np.array([[base[i][k] for k in keys] for i in range(n_it)])
    allv = np.array([[[r[i][k] for k in keys] for i in range(n_it)] for r in [base] + runs])
    lo, hi = np.nanmin(allv, axis=0), np.nanmax(allv, axis=0)
    port, _ = cpg_port()
    w0 = {k: v.detach().numpy().copy() for k, v in port.policy.params.items()}
    out = os.path.join(ROOT, "tests/golden", CPG["golden"] + ".npz")
    np.savez_compressed(out, meta=np.array(repr(dict(torch=torch.__version__, numpy=np.__version__, runs=1 + len(runs), variants=[os.path.basename(p) for p in others]))),
                        stream_seed=CPG["stream_seed"], N=CPG["N"], T=CPG["T"], seed=CPG["seed"], metric_keys=np.array(keys), base=val, lo=lo, hi=hi,
                        **{f"w0/{k}": v for k, v in w0.items()})
    print(f"wrote {out} ({os.path.getsize(out) / 1024:.0f} KB): {n_it} rollouts + updates, {len(keys)} scalars, {1 + len(runs)} runs of the CPU port\n")
    show = [k for k in ("train/nu", "train/average_cost", "train/early_stop_epoch", "train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss", "train/approx_kl",
                        "train/reward_explained_variance", "train/std") if k in keys]
    print("| rollout | " + " | ".join(show) + " |")
    print("|---|" + "---|" * len(show))
    for i in range(n_it):
        print(f"| {i} | " + " | ".join(f"{val[i, keys.index(k)]:.6g} [{lo[i, keys.index(k)]:.6g}, {hi[i, keys.index(k)]:.6g}]" for k in show) + " |")


SKIP = ("time/", "time(m)")


def band(base_path, others, name=None):
    base = json.load(open(base_path))["metrics"]
    runs = [json.load(open(p))["metrics"] for p in others]
    n_it = min([len(base)] + [len(r) for r in runs])
    keys = sorted(k for k in base[0] if not k.startswith(SKIP))
    val = np.array([[base[i][k] for k in keys] for i in range(n_it)])
    allv = np.array([[[r[i][k] for k in keys] for i in range(n_it)] for r in [base] + runs])
    lo, hi = np.nanmin(allv, axis=0), np.nanmax(allv, axis=0)
    w0, cn0 = base_init(name)
    out = os.path.join(ROOT, "tests/golden", ("g19_whole_run_hc" if name is None else CONFIGS[name]["golden"]) + ".npz")
    np.savez_compressed(out, meta=np.array(repr(dict(torch=torch.__version__, numpy=np.__version__, runs=1 + len(runs), variants=[os.path.basename(p) for p in others]))),
                        argv=np.array(ARGV if name is None else CONFIGS[name]["argv"]), stream_seed=STREAM_SEED if name is None else CONFIGS[name]["stream_seed"],
                        expert=np.array("tests/golden/expert_hc.npz" if name is None else CONFIGS[name]["expert"]),
                        uniform_streams=bool(name is not None and CONFIGS[name].get("uniform", False)), metric_keys=np.array(keys), base=val, lo=lo, hi=hi,
                        **{f"w0/{k}": v for k, v in w0.items()}, **{f"cn0/{k}": v for k, v in cn0.items()})
    print(f"wrote {out} ({os.path.getsize(out) / 1024:.0f} KB): {n_it} outer iterations, {len(keys)} metrics, {1 + len(runs)} runs of the CPU port\n")
    show = [k for k in ("forward/nu", "forward/average_cost", "true/cost", "true/reward", "forward/early_stop_epoch", "forward/reward_explained_variance",
                        "backward/cn_loss", "backward/kl_new_old", "backward/kl_old_new", "true/forward_kl", "true/reverse_kl") if k in keys]
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
    elif sys.argv[1] == "run2":
        run2(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else None)
    elif sys.argv[1] == "band2":
        band(sys.argv[3], sys.argv[4:], sys.argv[2])
    elif sys.argv[1] == "run3":
        run3(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else None)
    elif sys.argv[1] == "band3":
        band3(sys.argv[2], sys.argv[3:])
    else:
        band(sys.argv[2], sys.argv[3:])
