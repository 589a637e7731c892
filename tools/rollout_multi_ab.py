"""A/B of the multi-env rollout kernel at HC x 64 (one run): us per step for the kernel the environment selects
(MODE=auto|multi, ICRL_MULTI_E, ICRL_MULTI_PACK, ICRL_HIP_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
N, T = int(os.environ.get("N", "64")), int(os.environ.get("T", "2048"))
kind = os.environ.get("KIND", "hc")
od, ad = (18, 6) if kind == "hc" else (113, 8)
env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
lo = -np.ones(ad, np.float32)
cn = ConstraintNet(od, ad, [20] if kind == "hc" else [40, 40], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
env.set_cost_function(cn.cost_function)
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, seed=0)
agent._setup_learn(8 * N * T)
agent.rollout_kernel = os.environ.get("MODE", "multi")
ts = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
    torch.cuda.synchronize(); ts.append(time.time() - t0)
    agent.check_rollout_status()
print(f"{kind} N={N} {agent.rollout_kernel} E={os.environ.get('ICRL_MULTI_E', '-')} pack={os.environ.get('ICRL_MULTI_PACK', '0')}: " + " ".join(f"{1e6 * t / T:.2f}" for t in ts[1:]) + " us/step")
