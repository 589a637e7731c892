"""Does local-update data parallelism (mode A: every rank trains on its own env shard, ONE averaging all-reduce per outer iteration)
LEARN like one rank with the union of the envs?  2 ranks x 32 envs (gloo transport, both ranks on GPU 0) against 1 rank x 64 envs,
HCWithPos shapes, README.md:38 flags at n_steps 512, 10 outer iterations, same seed.  Prints per iteration nu, true/cost, true/reward,
forward/average_cost of rank 0 of the 2-rank job and of the 1-rank job.

    python tools/two_rank_learning.py            (parent: spawns the three child processes)
"""
import json, os, socket, subprocess, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITERS = int(os.environ.get("ITERS", "10"))


def child(envs):
    import numpy as np, torch
    from icrl_amd import distributed as D
    from icrl_amd.icrl import build_parser, outer_iteration, setup
    rank, world = D.init_from_env()
    expert = os.path.join(ROOT, "tests/golden/expert_hc.npz")
    argv = ["icrl", "-er", "10", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "10", "-ft", str(2 * 64 * 512 - 1),
            "-ni", str(ITERS), "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis",
            "-ctkno", "2.5", "-nt", str(envs), "--n_steps", "512", "-s", "0", "-v", "0", "--permutation", "device"]
    cfg = vars(build_parser().parse_args(argv))
    # the same number of env steps per outer iteration in both jobs: forward_timesteps counts THIS rank's steps
    cfg["forward_timesteps"] = 2 * envs * 512 - 1
    cfg.update(rank=rank, world_size=world)
    st = setup(types.SimpleNamespace(**cfg))
    rows = []
    for it in range(ITERS):
        m = outer_iteration(st, it)
        rows.append({k: float(m[k]) for k in ("forward/nu", "true/cost", "true/reward", "forward/average_cost", "backward/cn_loss")})
    if rank == 0:
        print("RESULT " + json.dumps(dict(world=world, envs=envs, rows=rows)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ICRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, __file__, "child", "32"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, __file__, "child", "64"], env=env1, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900).stdout.decode(errors="replace")
    res = {}
    for o in outs + [one]:
        for line in o.splitlines():
            if line.startswith("RESULT "):
                r = json.loads(line[7:]); res[r["world"]] = r["rows"]
    if 1 not in res or 2 not in res:
        print("\n".join(outs + [one])[-4000:]); raise SystemExit(1)
    print("iter |  nu (2 ranks x 32 | 1 rank x 64) | true/cost | true/reward | forward/average_cost")
    for it in range(ITERS):
        a, b = res[2][it], res[1][it]
        print(f"{it:4d} | {a['forward/nu']:.4f} {b['forward/nu']:.4f} | {a['true/cost']:.4f} {b['true/cost']:.4f} | {a['true/reward']:9.2f} {b['true/reward']:9.2f} | "
              f"{a['forward/average_cost']:.4f} {b['forward/average_cost']:.4f}")


if __name__ == "__main__":
    main()
