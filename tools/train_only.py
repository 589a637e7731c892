import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
kind, N, T = "hc", 64, 2048
od, ad = 18, 6
env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
lo = -np.ones(ad, np.float32)
cn = ConstraintNet(od, ad, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
env.set_cost_function(cn.cost_function)
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, n_epochs=int(os.environ.get("EPOCHS", "2")), seed=0, permutation="device")
agent._setup_learn(N * T)
agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
torch.cuda.synchronize(); t0 = time.time()
agent.train()
torch.cuda.synchronize(); print("train ms", 1e3 * (time.time() - t0))
agent.train_events = []
for variant in ("tiles", "auto", "tiles", "auto"):
    agent.train_kernel = variant
    agent.train_events.clear()
    agent.train()
    torch.cuda.synchronize()
    e0, e1, n = agent.train_events[-1]
    print(variant, "us/step", 1e3 * e0.elapsed_time(e1) / n, "steps", n)
for variant in ("tiles", "auto"):
    agent.train_kernel = variant
    agent.profile_phases = 1
    agent.train()
    torch.cuda.synchronize()
    st = agent._train_ws["stats"].cpu().numpy()
    print(variant, "cycles/step per phase (role 0 | 1 | 2[:6]):", np.round(st[12:19]), np.round(st[19:26]), np.round(st[26:32]))
agent.profile_phases = 0
