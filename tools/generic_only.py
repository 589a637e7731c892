"""us per optimiser step of the generic-shape update path (csrc/generic.hip): a policy with 128-wide layers / a 512-row batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
for name, kw, B in (("-pl 128 128 -rvl 128 128 -cvl 128 128, batch 64", dict(policy_kwargs=dict(net_arch=[dict(pi=[128, 128], vf=[128, 128], cvf=[128, 128])])), 64),
                    ("default widths, batch 256", {}, 256), ("default widths, batch 512", {}, 512),
                    ("-pl 128 128 -rvl 128 128 -cvl 128 128, batch 512", dict(policy_kwargs=dict(net_arch=[dict(pi=[128, 128], vf=[128, 128], cvf=[128, 128])])), 512),
                    ("-sl 64 -pl 128 128 -rvl 64 -cvl 64 64 64, batch 64", dict(policy_kwargs=dict(net_arch=[64, dict(pi=[128, 128], vf=[64], cvf=[64, 64, 64])])), 64))[slice(*[int(x) for x in os.environ.get("ONLY", "0,5").split(",")])]:
    N, T = 64, 256
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "hc", 0)))
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=2, seed=0, permutation="device", **kw)
    agent._setup_learn(3 * N * T)
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost"); torch.cuda.synchronize()      # (first call: module load)
    t0 = time.time()
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
    torch.cuda.synchronize(); t_roll = time.time() - t0
    t0 = time.time()
    agent._collect_rollouts_stepped(env, None, agent.rollout_buffer, T, "cost")
    torch.cuda.synchronize(); t_py = time.time() - t0
    agent.train(); torch.cuda.synchronize()
    t0 = time.time(); agent.train(); torch.cuda.synchronize(); dt = time.time() - t0
    steps = 2 * (N * T // B)
    if os.environ.get("PROF"):
        agent.profile_phases = 1; agent.train(); torch.cuda.synchronize(); agent.profile_phases = 0
        st_ = agent._train_ws["stats"].cpu().numpy()
        print("   cycles per step of workgroup 0: rows+stats | fwd | loss | bwd | wgrad | barrier A | reduce | barrier B | adam | barrier C:", np.round(st_[12:22]), "one XCD:", bool(st_[22]), "| of them: arrive B + rows | sums + operands | arrive C | rows commit | arrive A:", np.round(st_[23:28]))
    print(f"{name}: rollout {1e6 * t_roll / T:.0f} us per {N}-env step ({'rollout_generic_kernel: one persistent launch' if agent.policy.wide else 'fused'}; the Python loop over the fine-grained entry points: {1e6 * t_py / T:.0f}), update {1e6 * dt / steps:.1f} us per optimiser step ({steps} steps)")
