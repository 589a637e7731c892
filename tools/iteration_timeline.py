"""wall-clock timeline of one outer ICRL iteration (host side, with a device sync after each stage)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from icrl_amd import icrl as I, utils, logger
from icrl_amd.vec_env import sync_envs_normalization

cfg = bench.config2(4, 0, 0, 1)
st = I.setup(cfg)
I.outer_iteration(st, 0)
agent = st["agent"]
for it in range(1, 3):
    torch.cuda.synchronize(); t = [time.time()]
    def mark():
        torch.cuda.synchronize(); t.append(time.time())
    # forward step, split
    total = agent._setup_learn(cfg.forward_timesteps, True)
    while agent.num_timesteps < total:
        agent.collect_rollouts(agent.env, None, agent.rollout_buffer, agent.n_steps, "cost"); mark()
        agent.train(); mark()
    st["timesteps"] += agent.num_timesteps
    sync_envs_normalization(st["train_env"], st["sampling_env"])
    oo, o, a, r, l = utils.sample_from_agent(agent, st["sampling_env"], cfg.expert_rollouts); mark()
    bw = st["constraint_net"].train(cfg.backward_iters, oo, a, l, None, None, 1 - it / cfg.n_iters); mark()
    st["train_env"].set_cost_function(st["constraint_net"].cost_function)
    sync_envs_normalization(st["train_env"], st["eval_env"])
    utils.evaluate_policy(agent, st["eval_env"], n_eval_episodes=10, deterministic=False); mark()
    fk = utils.compute_kl(agent, st["d_expert_obs"], st["d_expert_acs"], st["expert_agent"])
    rk = utils.compute_kl(st["expert_agent"], oo, a, agent); mark()
    names = ["rollout", "train", "rollout", "train", "sample", "cn.train", "evaluate", "kl"]
    d = np.diff(t) * 1e3
    print("iteration", it, " ".join(f"{n}={x:.1f}" for n, x in zip(names, d)), f"total={d.sum():.1f} ms")
