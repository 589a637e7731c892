"""does the update step time depend on WHERE the exchange workspace lies in device memory?  (the step time is bimodal from process to
process: 8.86-8.91 or 9.13-9.15 us)  One process, the same agent and data, the `sync` workspace moved through a large buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icrl_amd.ppo_lag import PPOLagrangian
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
from icrl_amd.constraint_net import ConstraintNet
N, T, B = 64, 2048, 64
env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "hc", 0)))
lo = -np.ones(6, np.float32)
cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
env.set_cost_function(cn.cost_function)
agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=2, seed=0, permutation="device")
agent._setup_learn(N * T)
agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost")
agent.train(); torch.cuda.synchronize()
agent.train_events = []
n_words = agent._train_ws["sync"].numel()
big = torch.zeros(64 * 1024 * 1024 // 8 + n_words, dtype=torch.int64, device="cuda")
res = []
for off_kb in (list(range(0, 16384, 512)) if os.environ.get('FINE') else [0, 4, 64, 256, 1024, 2048, 3072, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536]):
    off = min(off_kb * 1024 // 8, big.numel() - n_words)
    agent._train_ws["sync"] = big[off:off + n_words]
    agent.train_events.clear(); agent.train(); torch.cuda.synchronize()
    e0, e1, n = agent.train_events[-1]
    res.append((off_kb, 1e3 * e0.elapsed_time(e1) / n, (agent._train_ws["sync"].data_ptr() >> 21) & 3))
print("update us/step by offset of the sync workspace (KB) [address bits 22:21]: " + ", ".join(f"{k}: {v:.3f} [{b}]" for k, v, b in res))
# separate fresh allocations: does the address bit decide?
fresh = []
keep = []
for i in range(12):
    t = torch.zeros(n_words + (i % 3) * 300000, dtype=torch.int64, device="cuda"); keep.append(t)
    agent._train_ws["sync"] = t[:n_words]
    agent.train_events.clear(); agent.train(); torch.cuda.synchronize()
    e0, e1, n = agent.train_events[-1]
    fresh.append((hex(t.data_ptr()), (t.data_ptr() >> 21) & 3, 1e3 * e0.elapsed_time(e1) / n))
print("fresh allocations (address, bits 22:21, us/step): " + ", ".join(f"{a} [{b}] {v:.3f}" for a, b, v in fresh))
# and the other suspects: fresh allocations of the policy parameters' neighbours do not move; re-run at offset 0
agent._train_ws["sync"] = big[:n_words]
agent.train_events.clear(); agent.train(); torch.cuda.synchronize()
e0, e1, n = agent.train_events[-1]
print(f"offset 0 again: {1e3 * e0.elapsed_time(e1) / n:.3f}")
