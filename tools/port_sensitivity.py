"""How touchy is a loop configuration?  Runs the CPU port (oracle/loop.py) twice — from the given initial weights and from the same
weights perturbed by a relative `--eps` — and prints, per PPO-Lagrangian update, the distance between the two runs' parameters.

A trajectory comparison product-vs-port can only say something about the kernels while the port compared with ITSELF stays smooth at
the size of the product's own drift (observations: ~2e-5 after four rollouts).  Used to pick the stream seed of
tests/test_icrl_trajectory_gpu.py::test_icrl_hc_shared_trunk_two_iterations_vs_port (seeds 12, 13, 15 jump to 1e-3 at eps 3e-5, 14 stays
at 5e-5).  CPU only; test infrastructure.

  python tools/port_sensitivity.py --eps 3e-5 --streams 14 [--init init.npz] -- -sl 48 -pl 64 32 32 -rvl 40 -cvl
(--init: npz with policy/<name> and cn/<name> arrays, e.g. dumped from the product's setup(); default: the port's own seeded init)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loop as o_loop                      # noqa: E402
from oracle.streams import SeededStreams               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eps", type=float, default=3e-5)
    ap.add_argument("--streams", type=int, default=12)
    ap.add_argument("--init", default=None)
    ap.add_argument("flags", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    from icrl_amd.icrl import build_parser
    expert = os.path.join(ROOT, "tests/golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "10", "-ft", "2000", "-ni", "2",
            "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis", "-ctkno", "2.5", "-nt", "8",
            "--n_steps", "128", "-s", "3", "-v", "0"] + [f for f in a.flags if f != "--"]
    cfg = vars(build_parser().parse_args(argv))
    cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    ex = np.load(expert)
    sub = lambda g, p: {k[len(p):]: g[k] for k in g.files if k.startswith(p)}
    run = lambda init, n_iters=None: o_loop.icrl_port(cfg, ex["observations"][:1000], ex["actions"][:1000], sub(ex, "policy/"),
                                                      streams=SeededStreams(a.streams), init=init, n_iters=n_iters)
    if a.init:
        g = np.load(a.init)
        base = dict(policy=sub(g, "policy/"), cn=sub(g, "cn/"))
    else:
        objs = run(None, 0)[3]
        base = dict(policy={k: v.detach().numpy().copy() for k, v in objs["agent"].policy.params.items()},
                    cn={k: v.detach().numpy().copy() for k, v in objs["cn"].state_dict().items()})
    snaps, orig = [], o_loop.ppo_lag_train

    def rec(policy, *args, **kw):
        out = orig(policy, *args, **kw)
        snaps[-1].append(({n: v.detach().numpy().copy() for n, v in policy.params.items()}, out.get("train/approx_kl")))
        return out
    o_loop.ppo_lag_train = rec
    for eps in (0.0, a.eps):
        snaps.append([])
        rs = np.random.RandomState(1)
        run(dict(policy={k: (v * (1 + eps * np.sign(rs.randn(*v.shape)))).astype(np.float32) for k, v in base["policy"].items()}, cn=base["cn"]))
    for i, (p, q) in enumerate(zip(*snaps)):
        w = {k: float(np.abs(p[0][k] - q[0][k]).max()) for k in p[0]}
        worst = max(w, key=w.get)
        print(f"update {i}: approx_kl {p[1]:.7g} vs {q[1]:.7g}; largest parameter distance {w[worst]:.3g} ({worst})")


if __name__ == "__main__":
    main()
