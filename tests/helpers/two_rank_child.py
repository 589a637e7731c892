"""Child process of tests/test_multirank_gpu.py: ONE rank of a 2-rank run (gloo transport, both ranks on GPU 0).
Phase A: one sharded rollout without observation normalisation + the collective -> the merged running moments must equal
         those of one process that stepped the union of the shards (checked by the parent).
Phase B: icrl setup() + 2 outer_iteration()s with the per-iteration collective; dumps what must agree across ranks.
Phase C: cpg (BASELINE configs[4]'s flags at toy sizes) — learn() of 3 rollouts + updates with callbacks.RankSyncCallback."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N, T, A = 8, 64, 6


def shard_noise(seed, k, rank):
    """rollout k of the job: [T, world*N, A] normals, this rank's columns."""
    full = np.random.RandomState([seed, k]).randn(T, 2 * N, A).astype(np.float32)
    return full[:, rank * N:(rank + 1) * N]


def main():
    out_path = sys.argv[1]
    from icrl_amd import distributed as D, utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.icrl import build_parser, outer_iteration, setup
    from icrl_amd.ppo_lag import PPOLagrangian
    rank, world = D.init_from_env()
    assert world == 2
    out = {}
    # ---------------- phase A
    env = utils.make_train_env("HCWithPos-v0", None, True, 5, N, normalize_obs=False, cost_info_str="cost", reward_gamma=0.99,
                               cost_gamma=0.99, env_index_offset=rank * N)
    lo = -np.ones(A, np.float32)
    torch.manual_seed(1)
    cn = ConstraintNet(18, A, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=64, n_epochs=2, seed=5)
    out["A_keys"] = env.unwrapped.key.cpu().numpy()
    out["A_params0"] = agent.policy.params.cpu().numpy()
    rms_list = [env.obs_rms, env.ret_rms, env.cost_rms]
    prev = [D.moments_to_sums(r.mean, r.var, r.count) for r in rms_list]
    agent._setup_learn(N * T)
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=torch.as_tensor(shard_noise(3, 0, rank), device="cuda"))
    agent.check_rollout_status()
    out["A_orig_obs"] = agent.rollout_buffer.new_orig_observations.cpu().numpy()
    out["A_rewards"] = agent.rollout_buffer.rewards.cpu().numpy()
    D.allreduce_state([], rms_list, prev, world)
    for name, r in zip(("obs", "ret", "cost"), rms_list):
        out[f"A_{name}_rms"] = np.concatenate([np.atleast_1d(r.mean), np.atleast_1d(r.var), [r.count]])
    # ---------------- phase B
    expert = os.path.join(ROOT, "tests/golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "--expert_agent_path", expert, "-tk", "0.01", "-cl", "20", "-bi", "3", "-ft", "1024",
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-crc", "0.5", "-psis", "-ctkno", "2.5",
            "-nt", str(N), "--n_steps", str(T), "-ne", "4", "-s", "9", "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=rank, world_size=world)
    st = setup(types.SimpleNamespace(**cfg))
    out["B_keys"] = st["train_env"].unwrapped.key.cpu().numpy()
    out["B_params0"] = st["agent"].policy.params.cpu().numpy()
    nus = []
    for it in range(2):
        m = outer_iteration(st, it)
        nus.append(m["forward/nu"])
        if it == 0:
            out["B_first_obs"] = st["agent"].rollout_buffer.orig_observations[0].cpu().numpy()
    pol, cnet, dual = st["agent"].policy, st["constraint_net"], st["agent"].dual
    out.update(B_params=pol.params.cpu().numpy(), B_exp_avg=pol.exp_avg.cpu().numpy(), B_exp_avg_sq=pol.exp_avg_sq.cpu().numpy(),
               B_cn=cnet.params.cpu().numpy(), B_cn_m=cnet.exp_avg.cpu().numpy(), B_cn_v=cnet.exp_avg_sq.cpu().numpy(),
               B_dual=np.array([dual.log_nu, dual.m, dual.v, dual.t], np.float64),
               B_steps=np.array([pol.adam_step, cnet.adam_step]), B_logged_nu=np.array(nus), B_timesteps=np.array([st["timesteps"]]))
    for name, r in zip(("obs", "ret", "cost"), st["rms_list"]):
        out[f"B_{name}_rms"] = np.concatenate([np.atleast_1d(r.mean), np.atleast_1d(r.var), [r.count]])
    # ---------------- phase C
    from icrl_amd import cpg as C
    cargv = ["cpg", "--cn_path", os.path.join(ROOT, "tests/golden/cn_antbroken.npz"), "-tei", "AntWallBroken-v0", "-eei", "AntWallBrokenTest-v0",
             "-tk", "0.01", "--batch_size", "128", "--reward_gae_lambda", "0.9", "--n_epochs", "3", "--learning_rate", "3e-5", "--clip_range", "0.4",
             "-t", str(3 * 16 * 64), "-plr", "1.0", "-nt", "16", "--n_steps", "64", "-s", "4", "-v", "0", "--eval_every_rollouts", "100"]
    ccfg = vars(C.build_parser().parse_args(cargv))
    ccfg.update(rank=rank, world_size=world, save_dir=None)
    model, hist = C.cpg(types.SimpleNamespace(**ccfg), log=None)
    cpol, cdual, cenv = model.policy, model.dual, model.env
    out.update(C_params=cpol.params.cpu().numpy(), C_exp_avg=cpol.exp_avg.cpu().numpy(), C_exp_avg_sq=cpol.exp_avg_sq.cpu().numpy(),
               C_dual=np.array([cdual.log_nu, cdual.m, cdual.v, cdual.t], np.float64), C_steps=np.array([cpol.adam_step]),
               C_keys=cenv.unwrapped.key.cpu().numpy(), C_timesteps=np.array([model.num_timesteps]),
               C_first_obs=model.rollout_buffer.orig_observations[0].cpu().numpy())
    for name, r in zip(("obs", "ret", "cost"), (cenv.obs_rms, cenv.ret_rms, cenv.cost_rms)):
        out[f"C_{name}_rms"] = np.concatenate([np.atleast_1d(r.mean), np.atleast_1d(r.var), [r.count]])
    np.savez(out_path, **out)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
