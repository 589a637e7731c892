import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
kind, N, T = "ant", 64, 512
od, ad = 113, 8
env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
lo = -np.ones(ad, np.float32)
cn = ConstraintNet(od, ad, [40, 40], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
env.set_cost_function(cn.cost_function)
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=128, n_epochs=4, seed=0, permutation="device")
agent._setup_learn(N * T)
agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
agent.train_events = []
for variant in ("tiles", "auto", "tiles", "auto"):
    agent.train_kernel = variant
    agent.train_events.clear(); agent.train(); torch.cuda.synchronize()
    e0, e1, n = agent.train_events[-1]
    print(variant, "ant B=128 us/step", 1e3 * e0.elapsed_time(e1) / n, "steps", n)
