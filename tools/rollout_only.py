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
    for mode in ("auto", "steps", "auto", "steps"):
        agent.rollout_kernel = mode
        torch.cuda.synchronize(); t0 = time.time()
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        torch.cuda.synchronize(); dt = time.time() - t0
        agent.check_rollout_status()
        print(f"{kind}{'-broken' if broken else ''} N={N} T={T} {mode:5s}: {1e3 * dt:8.2f} ms = {1e6 * dt / T:7.2f} us/step = {N * T / dt / 1e6:6.2f} M env-steps/s")

for cfg in (("hc", 64, 2048), ("hc", 256, 1024), ("ant", 256, 512), ("ant", 512, 256, True)):
    run(*cfg)
