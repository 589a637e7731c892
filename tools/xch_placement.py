"""does the persistent rollout's step time depend on where its exchange workspace lies (2 MB granularity)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
for kind, N, T in (("hc", 64, 2048), ("hc", 256, 1024), ("ant", 256, 512)):
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
    lo = -np.ones(ad, np.float32)
    cn = ConstraintNet(od, ad, [20] if kind == "hc" else [40, 40], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, seed=0)
    agent._setup_learn(N * T)
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
    n_words = agent._ag["xch_ws"].numel()
    big = torch.zeros(16 * 1024 * 1024 // 8 + n_words, dtype=agent._ag["xch_ws"].dtype, device="cuda")
    res = []
    for off_kb in range(0, 8192, 1024):
        agent._ag["xch_ws"] = big[off_kb * 128:off_kb * 128 + n_words]
        torch.cuda.synchronize(); t0 = time.time()
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        torch.cuda.synchronize(); res.append((off_kb, 1e6 * (time.time() - t0) / T))
    print(f"{kind} x {N}: rollout us/step by offset of the exchange workspace (KB): " + ", ".join(f"{k}: {v:.2f}" for k, v in res), flush=True)
