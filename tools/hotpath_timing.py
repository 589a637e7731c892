"""Quick timing of the forward step pieces at BASELINE config-2 / config-3 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet

def run(kind, N, T, B, E, lr, cl):
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
    lo = -np.ones(ad, np.float32)
    cn = ConstraintNet(od, ad, cl, None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=E, target_kl=None, learning_rate=lr, seed=0,
                          permutation="device")
    agent._setup_learn(N * T)
    agent.profile_phases = int(os.environ.get("PHASES", "0"))
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
        torch.cuda.synchronize(); t1 = time.time()
        agent.train()
        torch.cuda.synchronize(); t2 = time.time()
        steps = E * ((N * T + B - 1) // B)
        print(f"{kind} N={N} T={T} B={B} E={E}: rollout {1e3*(t1-t0):8.1f} ms ({1e6*(t1-t0)/T:6.1f} us/step)  "
              f"train {1e3*(t2-t1):8.1f} ms ({1e6*(t2-t1)/steps:6.2f} us/opt-step, {steps} steps)  -> {N*T/(t2-t0):10.0f} env-steps/s")
        if agent.profile_phases:
            pc = agent.phase_cycles
            names = ["fwd", "loss", "bwd", "norm+pub", "stage", "wait", "adam"]
            for role in range(2):
                print("   role", role, "cycles/step:", " ".join(f"{n}={pc[7*role+k]:.0f}" for k, n in enumerate(names)), " total", pc[7*role:7*role+7].sum())

run("hc", 64, 2048, 64, 10, 3e-4, [20])
run("ant", 256, 2048, 128, 20, 3e-5, [40, 40])
