"""CPU only: how far do two runs of the SAME algorithm drift apart over the full-size forward step of BASELINE configs[1]?

The full-size parity test (tests/test_fullsize_parity_gpu.py) compares the HIP path with the CPU port after 20 480 and 40 960 dependent
optimiser steps.  Its bounds must not be fitted to what the HIP path happened to show: this tool runs the PORT against ITSELF through the
same schedule (64 envs x 2048 steps, 10 epochs x 2048 minibatches of 64, target_kl 0.01, SeededStreams(77), seed 0) with a rounding-size
disturbance, and reports the same quantities the test asserts on.  Any correct fp32 implementation that differs from the port in
summation order is such a disturbance, so these numbers are the floor a HIP-vs-port bound can be held to.

    python tools/calibrate_drift.py run <variant> <out.json> [configs1|configs2]
                 variant: base | ulp / ulpm (every initial parameter moved to the next float32 above / below) | ulpc (the critics' only)
                          | ulp1 (ONE parameter moved by one ulp) | threads8 | rnd<k> (every parameter moved by -1 / 0 / +1 ulp, RandomState(k))
                          | rev / rot<k> (the rows of EVERY minibatch reversed / rotated by k: the same minibatches, another summation order in
                            every reduction of every optimiser step — the kind of difference a second implementation has)
                          | tanhe6<phase> (round 6: every tanh evaluation carries a 1e-6 relative pseudo-random error — the documented size of the kernels'
                            v_exp-based tanh against libm's: THE disturbance that is representative of the HIP path on schedules where single samples
                            crossing the clip boundary dominate the drift — configs2full / configs4shard)
                          | tanhx / tanhx2 (round 6: torch.tanh replaced by an exp-based float32 formula in every activation, forward and backward
                            — the other difference a second implementation has: its transcendental functions)
                 schedule: configs1 (default; test_configs1_forward_step_full_size) | configs2 (test_configs2_widths_long_chain: AntWall flags,
                          256 envs x 128 steps, batch 128, 20 epochs, lr 3e-5, seed 3)
    python tools/calibrate_drift.py compare base.json other.json ...      (the schedule is read from the files)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


SCHEDULES = {      # the arguments of tests/test_fullsize_parity_gpu.py's two tests
    "configs1": dict(kind="hc", N=64, T=2048, od=18, ad=6, cn=[20], seed=0, lr=3e-4,
                     kw=dict(batch_size=64, n_epochs=10, target_kl=0.01, penalty_learning_rate=0.1)),
    "configs2": dict(kind="ant", N=256, T=128, od=113, ad=8, cn=[40, 40], seed=3, lr=3e-5,
                     kw=dict(batch_size=128, n_epochs=20, target_kl=0.02, learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, cost_gae_lambda=0.9,
                             penalty_initial_value=0.1, penalty_learning_rate=0.05)),
    # round 6: the FULL row counts of BASELINE configs[2] and of the per-GPU shards of configs[3] / configs[4] (tests/test_fullsize_parity_gpu.py:
    # test_configs2_full_rows, test_configs3_shard_full_rows, test_configs4_shard_full_rows) — one rollout + one train() each
    "configs2full": dict(kind="ant", N=256, T=2048, od=113, ad=8, cn=[40, 40], seed=3, lr=3e-5, rollouts=1,
                         kw=dict(batch_size=128, n_epochs=20, target_kl=0.02, learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9, cost_gae_lambda=0.9,
                                 penalty_initial_value=0.1, penalty_learning_rate=0.05)),
    "configs3shard": dict(kind="hc", N=256, T=2048, od=18, ad=6, cn=[20], seed=0, lr=3e-4, rollouts=1,
                          kw=dict(batch_size=64, n_epochs=10, target_kl=0.01, penalty_learning_rate=0.1)),
    "configs4shard": dict(kind="ant", broken=True, N=512, T=2048, od=113, ad=8, cn="tests/golden/cn_antbroken.npz", seed=4, lr=3e-5, rollouts=1,
                          kw=dict(batch_size=128, n_epochs=20, target_kl=0.01, learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9,
                                  penalty_learning_rate=1.0)),
}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_cost_net(sc):
    """the schedule's constraint net on the CPU: a seeded fresh one, or the reference's AntBroken checkpoint through its off-by-one load()
    (no clipping, no normalisation: constraint_net.py:394-399)."""
    from oracle import nets as o_nets
    od, ad = sc["od"], sc["ad"]
    if isinstance(sc["cn"], str):
        z = np.load(os.path.join(ROOT, sc["cn"]))
        ocn = o_nets.CostNet(od, ad, [int(h) for h in z["hidden_sizes"]], False, None, None, None, None, None)
        ocn.load_state_dict({k[len("cn_network/"):]: z[k] for k in z.files if k.startswith("cn_network/")})
        return ocn
    lo = -np.ones(ad, np.float32)
    torch.manual_seed(sc["seed"] + 1)
    return o_nets.CostNet(od, ad, sc["cn"], False, None, None, 20, lo, -lo)


def run(variant, out_path, schedule="configs1"):
    from oracle import loop as o_loop, nets as o_nets
    from oracle.streams import SeededStreams
    torch.set_num_threads(8 if variant == "threads8" else 1)
    if variant.startswith("tanhx"):      # another tanh in EVERY activation, forward and backward: 1 - 2 / (exp(2 x) + 1) (tanhx) or (e^x - e^-x) / (e^x + e^-x) (tanhx2) in
        f = (lambda x: 1 - 2 / (torch.exp(2 * x) + 1)) if variant == "tanhx" else (lambda x: (torch.exp(x) - torch.exp(-x)) / (torch.exp(x) + torch.exp(-x)))
        torch.tanh = f                   # float32 instead of libm's — it turns out as accurate as libm's (~1e-7 relative): a weak disturbance
    if variant.startswith("tanhe6"):     # tanh with a 1e-6-RELATIVE pseudo-random error in every evaluation (deterministic in x; phase = the variant's suffix): the size of
        orig, ph = torch.tanh, float(variant[6:] or 0)      # the difference the kernels' v_exp_f32-based tanh has from libm's ("~1e-6 relative in every evaluation")
        torch.tanh = lambda x: orig(x) * (1 + 1e-6 * torch.sin(12345.678 * x + ph))
    sc = SCHEDULES[schedule]
    N, T, od, ad, seed = sc["N"], sc["T"], sc["od"], sc["ad"], sc["seed"]
    ocn = make_cost_net(sc)
    stack = o_loop.make_stack(N, sc["kind"], seed, broken=sc.get("broken", False)); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, seed=seed, **sc["kw"])
    n_epochs = sc["kw"]["n_epochs"]
    with torch.no_grad():
        if variant.startswith("rnd"):      # every parameter independently one ulp down / unchanged / one ulp up
            rng = np.random.RandomState(int(variant[3:]))
            for p in port.policy.parameters():
                m = torch.as_tensor(rng.randint(-1, 2, size=tuple(p.shape)))
                up = torch.nextafter(p, torch.full_like(p, float("inf"))); dn = torch.nextafter(p, torch.full_like(p, float("-inf")))
                p.copy_(torch.where(m > 0, up, torch.where(m < 0, dn, p)))
        if variant == "ulp":
            for p in port.policy.parameters():
                p.copy_(torch.nextafter(p, torch.full_like(p, float("inf"))))
        elif variant == "ulpm":       # every parameter to the next float32 below
            for p in port.policy.parameters():
                p.copy_(torch.nextafter(p, torch.full_like(p, float("-inf"))))
        elif variant == "ulpc":       # the critics' parameters only (the policy network starts identical)
            for n, p in port.policy.params.items():
                if "value" in n or "vf" in n:
                    p.copy_(torch.nextafter(p, torch.full_like(p, float("inf"))))
        elif variant == "ulp1":
            p = next(iter(port.policy.parameters()))
            p.view(-1)[0] = torch.nextafter(p.view(-1)[0], torch.tensor(float("inf")))
    streams = SeededStreams(77)
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    res, t0 = [], time.time()
    for k in range(sc.get("rollouts", 2)):
        b = port.collect_rollouts(streams.rollout_noise(T, N, port.act_dim))
        B = sc["kw"]["batch_size"]

        def perm_of(e):
            p = np.asarray(streams.permutation(e, T * N))
            if variant == "rev":
                p = p.reshape(-1, B)[:, ::-1].reshape(-1).copy()
            elif variant.startswith("rot"):
                p = np.roll(p.reshape(-1, B), int(variant[3:]), axis=1).reshape(-1).copy()
            return p
        out = port.train(perm_of)
        streams.consumed(min(int(out["train/early_stop_epoch"]) + 1, n_epochs))
        res.append(dict(
            schedule=schedule, steps=int(next(iter(port.optimizer.state.values()))["step"]),
            scalars={k_: float(out[k_]) for k_ in ("train/nu", "train/average_cost", "train/early_stop_epoch", "train/policy_gradient_loss",
                                                   "train/reward_value_loss", "train/cost_value_loss", "train/approx_kl", "train/clip_fraction")},
            epoch_kls=[float(x) for x in out.get("epoch_kls", [])],
            params={n: v.detach().numpy().ravel().tolist() for n, v in port.policy.params.items()},
            buffer={f: getattr(b, f).astype(np.float64).ravel()[::17].tolist() for f in ("rewards", "costs", "reward_values", "reward_advantages", "cost_advantages", "log_probs")}))
        print(variant, "train", k + 1, "done", round(time.time() - t0, 1), "s", flush=True)
    json.dump(res, open(out_path, "w"))


def compare(base_path, others):
    base = json.load(open(base_path))
    lr = SCHEDULES[base[0].get("schedule", "configs1")]["lr"]
    print("| disturbance | after | nu | average_cost | losses pg / rv / cv | approx_kl | clip_fraction | early_stop_epoch | max abs parameter difference | buffer (sampled) values / advantages |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    rates = [[], []]
    worst = [dict(), dict()]
    for path in others:
        o = json.load(open(path))
        for k in range(len(base)):
            a, b = base[k], o[k]
            d = lambda key: abs(a["scalars"][key] - b["scalars"][key])
            dp = max(float(np.abs(np.asarray(a["params"][n]) - np.asarray(b["params"][n])).max()) for n in a["params"])
            steps = a.get("steps") or 20480 * (k + 1)
            db = {f: float(np.abs(np.asarray(a["buffer"][f]) - np.asarray(b["buffer"][f])).max()) for f in a["buffer"]}
            rates[k].append(dp / (lr * steps))
            for key in ("train/nu", "train/average_cost", "train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss", "train/approx_kl", "train/clip_fraction"):
                worst[k][key] = max(worst[k].get(key, 0.0), d(key))
            print(f"| {os.path.basename(path).replace('.json', '')} | {steps} steps | {d('train/nu'):.1e} | {d('train/average_cost'):.1e} | "
                  f"{d('train/policy_gradient_loss'):.1e} / {d('train/reward_value_loss'):.1e} / {d('train/cost_value_loss'):.1e} | {d('train/approx_kl'):.1e} | "
                  f"{d('train/clip_fraction'):.1e} | {int(a['scalars']['train/early_stop_epoch'])}, {int(b['scalars']['train/early_stop_epoch'])} | "
                  f"{dp:.2e} = {dp / (lr * steps):.1e} x lr x steps | {db['reward_values']:.1e} / {db['reward_advantages']:.1e} |")
    print()
    for k in range(len(base)):
        r = np.sort(np.asarray(rates[k]))
        print(f"after train() #{k + 1} ({len(r)} disturbances): max |d param| / (lr x steps): min {r[0]:.2e}, median {np.median(r):.2e}, "
              f"90th percentile {np.percentile(r, 90):.2e}, max {r[-1]:.2e}; sorted: " + " ".join(f"{x:.2e}" for x in r))
        print("   worst logged differences: " + ", ".join(f"{key.split('/')[1]} {v:.1e}" for key, v in worst[k].items()))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "configs1")
    else:
        compare(sys.argv[2], sys.argv[3:])
