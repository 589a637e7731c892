"""rollout timing: persistent launch vs the per-step launch pairs, at HC x 64 / HC x 256 / AntWall x 256 / AntWallBroken x 512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet

def run(kind, N, T, broken=False):
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0, broken=broken)))
    lo = -np.ones(ad, np.float32)
    cn = ConstraintNet(od, ad, [20] if kind == "hc" else [40, 40], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, seed=0)
    agent._setup_learn(N * T)
    for mode in os.environ.get("MODES", "auto,steps,auto,steps").split(","):
        agent.rollout_kernel = mode
        torch.cuda.synchronize(); t0 = time.time()
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        torch.cuda.synchronize(); dt = time.time() - t0
        agent.check_rollout_status()
        print(f"{kind}{'-broken' if broken else ''} N={N} T={T} {mode:5s}: {1e3 * dt:8.2f} ms = {1e6 * dt / T:7.2f} us/step = {N * T / dt / 1e6:6.2f} M env-steps/s")
    if N <= 128 and os.environ.get("PHASES"):         # phase timers of the one-workgroup-per-env persistent kernel (workgroup 0, thread 0)
        import ctypes
        from icrl_amd import _lib
        agent.rollout_kernel = "auto"; agent.profile_phases = 1
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        _lib.lib().icrl_debug_rollout_profile(out)
        Tn = max(1, out[3])
        names = {0: "policy+env+publish", 1: "granule wait", 4: "statistics A", 5: "statistics B", 6: "normalise", 2: "rest"}
        print("   cycles/step: " + ", ".join(f"{v} {out[k] / Tn:.0f}" for k, v in names.items()) + f"  (sum {sum(out[k] for k in names) / Tn:.0f}); poll rounds of thread 0 per step {out[7] / Tn:.1f}")
        tr = (ctypes.c_ulonglong * (4 * N))()
        _lib.lib().icrl_debug_rollout_trace_wide(tr, N)
        tr = (np.array(list(tr), dtype=np.float64).reshape(N, 4) - min(tr)) / 100.0
        print("   one step, end of each wave's env-phase part, us after the first (min / median / max over workgroups): " +
              ", ".join(f"wave {k}: {tr[:, k].min():.2f} / {np.median(tr[:, k]):.2f} / {tr[:, k].max():.2f}" for k in range(4)))
        agent.profile_phases = 0
    if os.environ.get("PHASES_MULTI"):        # phase timers of the multi-env kernel (cycles per step; wave 0 and the cost wave)
        import ctypes
        from icrl_amd import _lib
        agent.rollout_kernel = "multi"
        for flag, who in ((1, "workgroup 0"), (2, "last workgroup")):
            agent.profile_phases = 1
            agent._wide_prof_flag = flag
            agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
            torch.cuda.synchronize()
            out = (ctypes.c_ulonglong * 16)()
            _lib.lib().icrl_debug_rollout_profile_wide(out)
            names = ("MLPs+heads", "env steps / cost net + rows", "owner gather+moments+publish", "statistics wait", "normalise+rows")
            for base, wave in ((0, "wave 0"), (8, "wave 3 (cost)")):
                Tn = max(1, out[base + 5])
                print(f"   multi, {who}, {wave}: " + ", ".join(f"{n} {out[base + k] / Tn:.0f}" for k, n in enumerate(names)) + f"  (sum {sum(out[base:base + 5]) / Tn:.0f} cycles/step" + (f"; cost wave: inputs prepared after {out[base + 6] / Tn:.0f}, hidden layers after {out[base + 7] / Tn:.0f}" if base else "") + ")")
        agent.profile_phases = 0
        agent.rollout_kernel = "auto"
    if N > 128 and os.environ.get("PHASES"):          # phase timers of the many-environment persistent kernel (cycles per step)
        import ctypes
        from icrl_amd import _lib
        agent.rollout_kernel = "auto"
        for flag, who in ((1, "workgroup 0 (env + column owner)"), (2, "last workgroup (env only)")):
            agent.profile_phases = 1
            agent._wide_prof_flag = flag
            agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
            torch.cuda.synchronize()
            out = (ctypes.c_ulonglong * 16)()
            _lib.lib().icrl_debug_rollout_profile_wide(out)
            names = ("policy+env+rows", "owner gather", "owner moments+publish", "statistics wait", "normalise+rows")
            for base, wave in ((0, "wave 0"), (8, "wave 1")):
                Tn = max(1, out[base + 5])
                print(f"   {who}, {wave}: " + ", ".join(f"{n} {out[base + k] / Tn:.0f}" for k, n in enumerate(names)) + f"  (sum {sum(out[base:base + 5]) / Tn:.0f} cycles/step)")
        G = min(N, 256)
        tr = (ctypes.c_ulonglong * (4 * G))()
        _lib.lib().icrl_debug_rollout_trace_wide(tr, G)
        tr = np.array(list(tr), dtype=np.float64).reshape(G, 4)
        t0 = tr[:, 0].min()
        us = lambda x: (x - t0) / 100.0
        owners = tr[:, 2] > 0
        print(f"   one step, us after the first workgroup left its env phase: env phase ends {us(tr[:, 0]).min():.1f}..{us(tr[:, 0]).max():.1f}; "
              f"owners' gathers end {us(tr[owners, 1]).min():.1f}..{us(tr[owners, 1]).max():.1f}; publishes {us(tr[owners, 2]).min():.1f}..{us(tr[owners, 2]).max():.1f} "
              f"(latest: workgroups {np.argsort(-tr[:, 2])[:4].tolist()}, {np.sort(us(tr[:, 2]))[-4:][::-1].round(1).tolist()}); statistics read {us(tr[:, 3]).min():.1f}..{us(tr[:, 3]).max():.1f}")
        agent.profile_phases = 0

CFGS = dict(hc64=("hc", 64, 2048), hc256=("hc", 256, 1024), ant256=("ant", 256, 512), antb512=("ant", 512, 256, True), hc128=("hc", 128, 1024), hc16=("hc", 16, 2048))
for name in os.environ.get("CFGS", "hc64,hc256,ant256,antb512").split(","):
    run(*CFGS[name])
