import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd import _lib as _l
if os.environ.get("ICRL_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["ICRL_LIB"])       # tools only: a variant build (e.g. -DICRL_FINE_PROF)
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
kind = os.environ.get("KIND", "hc")
N, T, B = (64, 2048, 64) if kind == "hc" else (64, 512, 128)
od, ad = (18, 6) if kind == "hc" else (113, 8)
env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 0)))
lo = -np.ones(ad, np.float32)
cn = ConstraintNet(od, ad, [20] if kind == "hc" else [40, 40], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
env.set_cost_function(cn.cost_function)
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=int(os.environ.get("EPOCHS", "2")), seed=0, permutation="device")
if os.environ.get("TUNE") == "0":
    agent.tune_sync_placement = False       # (A/B: no placement calibration of the exchange workspace)
agent._setup_learn(N * T)
agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
torch.cuda.synchronize(); t0 = time.time()
agent.train()
torch.cuda.synchronize(); print("train ms", 1e3 * (time.time() - t0))
agent.train_events = []
variants = os.environ.get("VARIANTS", "rows,auto,rows,auto").split(",")
for variant in variants:
    agent.train_kernel = variant
    agent.train_events.clear()
    agent.train()
    torch.cuda.synchronize()
    e0, e1, n = agent.train_events[-1]
    print(kind, variant, "us/step", round(1e3 * e0.elapsed_time(e1) / n, 3), "steps", n)
for variant in sorted(set(v for v in variants if v != "rows1")):
    agent.train_kernel = variant
    agent.profile_phases = 1
    agent.train()
    torch.cuda.synchronize()
    st = agent._train_ws["stats"].cpu().numpy()
    print(variant, "cycles/step per phase fwd|loss|bwd|wgrad|norm|wait|adam (role 0 | 1):", np.round(st[12:19]), np.round(st[19:26]))
agent.profile_phases = 0
if os.environ.get("FINE"):
    # per-wave phase timers of a -DICRL_FINE_PROF build (tools/build_variant.sh fine -DICRL_FINE_PROF; ICRL_LIB=...): FINE=1 all
    # eight waves of the policy workgroup, FINE=<role> those of another role
    names = ["L1", "prefetch", "S1", "L2", "head", "S3", "loss", "dH2", "S4", "dH1", "dW2", "dWh", "S5", "dW1", "norm", "staging", "poll+S6", "Adam", "S7", "S7a"]
    frole = int(os.environ["FINE"]) if os.environ["FINE"] in "012" else 0
    agent.train_kernel = "auto"
    print("role", frole, "wave " + " ".join(f"{n:>7s}" for n in names[:20]) + "   total")
    for wv in range(8):
        agent.profile_phases = 1 | (wv << 8) | (frole << 12); agent.train(); torch.cuda.synchronize()
        st = agent._train_ws["stats"].cpu().numpy()
        print(f"        {wv}    " + " ".join(f"{v:7.0f}" for v in st[12:32]), f"  {st[12:32].sum():.0f}")
