"""CPU only: how far do two fp32 executions of the reference algorithm drift apart over BASELINE configs[0] at FULL size (LGW-v0, 1 env,
n_steps 2000, README.md:25 flags, 2 outer iterations)?  Discrete actions are drawn through the inverse CDF of teacher-forced uniforms,
so a probability that differs in the 4th digit after ~8 000 dependent Adam steps flips an action now and then, and everything
downstream of the flipped sample (the episode, the constraint-net update fed with it, the next forward step) moves.  This tool runs
the CPU port against ITSELF with every initial policy parameter moved by -1 / 0 / +1 float32 ulp (RandomState(k)) and prints the
per-metric differences; tests/test_icrl_trajectory_gpu.py::test_icrl_lgw_full_size_two_iterations_vs_port takes its bounds from it.

    python tools/calibrate_lgw.py run <k|base> out.json        python tools/calibrate_lgw.py compare base.json other.json ...
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(variant, out):
    from oracle import loop as o_loop, nets as o_nets
    from oracle.streams import SeededStreams
    torch.set_num_threads(1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex = np.load(os.path.join(root, "tests/golden/expert_lgw.npz"))
    esd = {k[len("policy/"):]: ex[k] for k in ex.files if k.startswith("policy/")}
    cfg = dict(train_env_id="LGW-v0", eval_env_id="CLGW-v0", num_threads=1, seed=0, n_steps=2000, target_kl=0.01, cn_layers=(20,), cn_learning_rate=0.003,
               forward_timesteps=50000, n_iters=2, backward_iters=20, dont_normalize_obs=True, dont_normalize_reward=True, dont_normalize_cost=True,
               expert_rollouts=20)
    torch.manual_seed(0)
    pol = o_nets.TwoCriticPolicy(1, 2, discrete=True)
    sd = {k: v.detach().clone() for k, v in pol.state_dict().items()} if hasattr(pol, "state_dict") else None
    if variant != "base":
        rng = np.random.RandomState(int(variant))
        for k, p in sd.items():
            m = torch.as_tensor(rng.randint(-1, 2, size=tuple(p.shape)))
            up = torch.nextafter(p, torch.full_like(p, float("inf"))); dn = torch.nextafter(p, torch.full_like(p, float("-inf")))
            sd[k] = torch.where(m > 0, up, torch.where(m < 0, dn, p))
    torch.manual_seed(1)
    cn = o_nets.CostNet(1, 2, [20], True, None, None, 20, None, None)
    init = dict(policy={k: v.numpy() for k, v in sd.items()}, cn={k: v.detach().numpy() for k, v in cn.state_dict().items()})
    m, steps, dt, _ = o_loop.icrl_port(cfg, ex["observations"], ex["actions"], esd, streams=SeededStreams(5), n_iters=2, init=init)
    json.dump([{k: float(v) for k, v in mm.items()} for mm in m], open(out, "w"))
    print(variant, "done", round(dt, 1), "s")


def compare(base, others):
    b = json.load(open(base))
    keys = sorted(b[0])
    worst = [{}, {}]
    for path in others:
        o = json.load(open(path))
        for it in range(2):
            for k in keys:
                if np.isfinite(b[it][k]) and np.isfinite(o[it][k]):
                    worst[it][k] = max(worst[it].get(k, 0.0), abs(b[it][k] - o[it][k]))
    for it in range(2):
        print(f"iteration {it}: worst |difference| over {len(others)} disturbed runs (value of the undisturbed run in brackets)")
        for k in keys:
            if worst[it].get(k, 0.0) > 0:
                print(f"   {k:34s} {worst[it][k]:.3e}   [{b[it][k]:.6g}]")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
    else:
        compare(sys.argv[2], sys.argv[3:])
