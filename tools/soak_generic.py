"""Soak of the persistent generic-shape kernels: R rounds of (rollout + update) on two identically seeded agents — every round must leave bit-identical
parameters, Adam moments and buffers on both (races between workgroups would show as a difference, a lost wake-up as a time-out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
R = int(os.environ.get("ROUNDS", "40"))
for arch, B in (([64, dict(pi=[128, 128], vf=[64], cvf=[64, 64, 64])], 64), ([dict(pi=[128, 128], vf=[128, 128], cvf=[128, 128])], 144), ([48, dict(pi=[64, 32, 32], vf=[40], cvf=[])], 320)):
    agents = []
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)      # ONE constraint net for both agents (its initialisation is not seeded)
    for rep in range(2):
        N, T = 64, 64
        env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "hc", 0)))
        env.set_cost_function(cn.cost_function)
        a = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=3, seed=0, permutation="device", policy_kwargs=dict(net_arch=arch))
        a._setup_learn(R * N * T)
        agents.append((a, env, cn))
    t0 = time.time()
    for r in range(R):
        rng = (torch.get_rng_state(), torch.cuda.get_rng_state(), np.random.get_state())      # both agents draw the same noise / permutations
        for a, env, _ in agents:
            torch.set_rng_state(rng[0]); torch.cuda.set_rng_state(rng[1]); np.random.set_state(rng[2])
            a.collect_rollouts(env, None, a.rollout_buffer, 64, "cost")
            a.train()
        sa, sb = agents[0][0].policy.state_dict(), agents[1][0].policy.state_dict()
        same = all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(agents[0][0].policy.exp_avg, agents[1][0].policy.exp_avg) and \
            torch.equal(agents[0][0].rollout_buffer.observations, agents[1][0].rollout_buffer.observations)
        if not same:
            print("DIFFERENCE at round", r, arch, B, [k for k in sa if not torch.equal(sa[k], sb[k])][:3], torch.equal(agents[0][0].rollout_buffer.observations, agents[1][0].rollout_buffer.observations), torch.equal(agents[0][0].rollout_buffer.actions, agents[1][0].rollout_buffer.actions)); sys.exit(1)
    torch.cuda.synchronize()
    print(f"net_arch {arch}, batch {B}: {R} rounds x 2 agents identical, {time.time() - t0:.1f} s", flush=True)
print("soak ok")
