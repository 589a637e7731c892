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
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, n_epochs=10, seed=0, permutation="device")
agent._setup_learn(N * T)
for mode in ("steps", "auto", "steps", "auto"):
    agent.rollout_kernel = mode
    agent.profile_phases = 1 if os.environ.get("PHASES") else 0
    torch.cuda.synchronize(); t0 = time.time()
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
    torch.cuda.synchronize(); print(mode, "rollout ms", 1e3 * (time.time() - t0))
    if mode == "auto":
        import ctypes
        from icrl_amd import _lib
        out = (ctypes.c_ulonglong * 8)()
        f = _lib.lib().icrl_debug_rollout_profile
        if f is not None:
            f(out)
            print("  cycles/step: policy+env %.0f | barrier %.0f | rest %.0f || load exchange %.0f | statistics %.0f | normalise own %.0f" % tuple(out[i] / max(out[3], 1) for i in (0, 1, 2, 4, 5, 6)))
senv = utils_mod = None
from icrl_amd import utils as _u
senv = _u.make_eval_env("HCWithPos-v0", False, seed=0)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    _u.sample_from_agent(agent, senv, 10)
    torch.cuda.synchronize(); print("sample 10 episodes (parallel streams) ms", 1e3 * (time.time() - t0))
from icrl_amd import utils
eenv = utils.make_eval_env("HCWithPosTest-v0", False, seed=0)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    utils.evaluate_policy(agent, eenv, 10, deterministic=False)
    torch.cuda.synchronize(); print("eval 10 episodes ms", 1e3 * (time.time() - t0))
