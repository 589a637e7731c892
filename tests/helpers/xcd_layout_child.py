"""child of tests/test_ppo_train_gpu.py::test_run_major_layout_equals_packed_layout: one update at HC shapes (wave pairs), one at AntWall
shapes with batch 128 (four workgroups per network exchanging partial gradients, then the row-owning kernel's two) and one rollout through the multi-env kernel; prints a
digest of everything they leave.  Run once as is (a run's workgroups on one XCD, workgroup-scope granule stores) and once with
ICRL_NO_XCD_PACK=1 (run-major grids, agent-scope stores): the digests must agree."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icrl_amd.constraint_net import ConstraintNet            # noqa: E402
from icrl_amd.ppo_lag import PPOLagrangian                   # noqa: E402
from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost      # noqa: E402

h = hashlib.sha256()
for kind, N, T, B in (("hc", 16, 64, 64), ("ant", 16, 64, 128)):
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    np.random.seed(11); torch.manual_seed(11)          # (the constraint net's initial weights come from the process-wide generators)
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, 3)))
    lo = -np.ones(ad, np.float32)
    cn = ConstraintNet(od, ad, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    torch.manual_seed(5)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=B, n_epochs=3, seed=3, target_kl=None)
    agent._setup_learn(2 * N * T)
    noise = torch.as_tensor(np.random.RandomState(1).randn(T, N, ad).astype(np.float32), device="cuda")
    agent.rollout_kernel = "multi"
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=noise)
    agent.check_rollout_status()
    rb = agent.rollout_buffer
    for k in ("observations", "actions", "rewards", "costs", "reward_advantages", "cost_returns"):
        h.update(getattr(rb, k).cpu().numpy().tobytes())
    print("PIECE", kind, "rollout", h.hexdigest()[:12])
    perms = np.stack([np.random.RandomState(9 + e).permutation(N * T) for e in range(3)])
    agent.train(perms=perms)
    for t in (agent.policy.params, agent.policy.exp_avg, agent.policy.exp_avg_sq):
        h.update(t.cpu().numpy().tobytes())
    print("PIECE", kind, "update", h.hexdigest()[:12])
    if kind == "ant":      # the row-owning kernel's two-workgroup form behind the default (four workgroups per network since round 6)
        agent.train_kernel = "rows"
        agent.train(perms=perms)
        for t in (agent.policy.params, agent.policy.exp_avg, agent.policy.exp_avg_sq):
            h.update(t.cpu().numpy().tobytes())
        print("PIECE", kind, "update (rows)", h.hexdigest()[:12])
    h.update(np.asarray(env.obs_rms.mean).tobytes())
print("DIGEST", h.hexdigest())
