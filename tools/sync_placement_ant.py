"""AntWall update (two workgroups per network, gradient exchange through the sync workspace): step time vs workspace position."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
N, T = 64, 512
env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "ant", 0)))
lo = -np.ones(8, np.float32)
cn = ConstraintNet(113, 8, [40, 40], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
env.set_cost_function(cn.cost_function)
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=128, n_epochs=4, seed=0, permutation="device")
agent.tune_sync_placement = False
agent._setup_learn(N * T)
agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
agent.train(); torch.cuda.synchronize()
agent.train_events = []
n_words = agent._train_ws["sync_words"]
big = torch.zeros(16 * 1024 * 1024 // 8 + n_words, dtype=torch.int64, device="cuda")
res = []
for off_kb in range(0, 8192, 256):
    agent._train_ws["sync"] = big[off_kb * 128:off_kb * 128 + n_words]
    agent.train_events.clear(); agent.train(); torch.cuda.synchronize()
    e0, e1, n = agent.train_events[-1]
    res.append((off_kb, 1e3 * e0.elapsed_time(e1) / n))
print("AntWall B=128 update us/step by offset of the sync workspace (KB): " + ", ".join(f"{k}: {v:.2f}" for k, v in res))
