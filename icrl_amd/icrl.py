"""ICRL outer loop — `python run_me.py icrl <flags>` with the reference's flags, on the MI355X-native hot path.

ref: icrl/icrl.py:45-312 (icrl), :314-464 (main / flag set).  Per outer iteration:
  forward   nominal_agent.learn(forward_timesteps, cost_function="cost")      (icrl.py:207)  -> fused rollouts + PPO-Lag kernel
  sample    sync_envs_normalization + sample_from_agent on a 1-env copy        (icrl.py:216-218)
  backward  constraint_net.train(...)                                          (icrl.py:235)  -> icrl_cn_train
  metrics   true cost, evaluate_policy (10 episodes), forward / reverse KL     (icrl.py:243-252)
Out of scope: W&B, plotting, video, curiosity / shaping callbacks, the GAIL baseline.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

from . import distributed as D, logger, utils
from .constraint_net import ConstraintNet
from .ppo_lag import PPOLagrangian
from .true_constraint_net import get_true_cost_function, mean_cost, null_cost
from .vec_env import VecNormalize, sync_envs_normalization
from . import spaces


def setup(config):
    """everything of icrl() before the loop (ref: icrl/icrl.py:45-197): env stacks, expert data, constraint net, nominal agent."""
    rank, world = getattr(config, "rank", 0), getattr(config, "world_size", 1)
    dev = config.device if str(config.device).startswith("cuda") else "cuda"
    train_env = utils.make_train_env(env_id=config.train_env_id, save_dir=config.save_dir, use_cost_wrapper=True,
                                     base_seed=config.seed, num_threads=config.num_threads,
                                     normalize_obs=not config.dont_normalize_obs, normalize_reward=not config.dont_normalize_reward,
                                     normalize_cost=not config.dont_normalize_cost, cost_info_str=config.cost_info_str,
                                     reward_gamma=config.reward_gamma, cost_gamma=config.cost_gamma,
                                     env_index_offset=rank * config.num_threads, device=dev)
    sampling_env = utils.make_eval_env(env_id=config.train_env_id, use_cost_wrapper=False, normalize_obs=not config.dont_normalize_obs,
                                       seed=config.seed + rank * config.num_threads, device=dev)
    eval_env = utils.make_eval_env(env_id=config.eval_env_id, use_cost_wrapper=False, normalize_obs=not config.dont_normalize_obs,
                                   seed=config.seed + rank * config.num_threads, device=dev)
    is_discrete = isinstance(train_env.action_space, spaces.Discrete)
    obs_dim = train_env.observation_space.shape[0]
    acs_dim = train_env.action_space.n if is_discrete else train_env.action_space.shape[0]
    action_low, action_high = None, None
    if isinstance(sampling_env.action_space, spaces.Box):
        action_low, action_high = sampling_env.action_space.low, sampling_env.action_space.high

    (expert_obs, expert_acs), expert_mean_reward = utils.load_expert_data(config.expert_path, config.expert_rollouts)
    expert_agent = None
    if getattr(config, "expert_agent_path", None):
        expert_agent = utils.load_expert_agent(config.expert_agent_path, dev)

    cn_lr_schedule = lambda x: (config.anneal_clr_by_factor ** (config.n_iters * (1 - x))) * config.cn_learning_rate
    constraint_net = ConstraintNet(
        obs_dim, acs_dim, config.cn_layers, config.cn_batch_size, cn_lr_schedule, expert_obs, expert_acs, is_discrete,
        config.cn_reg_coeff, config.cn_obs_select_dim, config.cn_acs_select_dim,
        no_importance_sampling=config.no_importance_sampling, per_step_importance_sampling=config.per_step_importance_sampling,
        clip_obs=config.clip_obs, initial_obs_mean=None if not config.cn_normalize else np.zeros(obs_dim),
        initial_obs_var=None if not config.cn_normalize else np.ones(obs_dim), action_low=action_low, action_high=action_high,
        target_kl_old_new=config.cn_target_kl_old_new, target_kl_new_old=config.cn_target_kl_new_old,
        train_gail_lambda=config.train_gail_lambda, eps=config.cn_eps, device=dev)
    train_env.set_cost_function(constraint_net.cost_function)

    def create_nominal_agent():
        agent = PPOLagrangian(
            policy=config.policy_name, env=train_env, learning_rate=config.learning_rate, n_steps=config.n_steps,
            batch_size=config.batch_size, n_epochs=config.n_epochs, reward_gamma=config.reward_gamma,
            reward_gae_lambda=config.reward_gae_lambda, cost_gamma=config.cost_gamma, cost_gae_lambda=config.cost_gae_lambda,
            clip_range=config.clip_range, clip_range_reward_vf=config.clip_range_reward_vf, clip_range_cost_vf=config.clip_range_cost_vf,
            ent_coef=config.ent_coef, reward_vf_coef=config.reward_vf_coef, cost_vf_coef=config.cost_vf_coef,
            max_grad_norm=config.max_grad_norm, use_sde=config.use_sde, sde_sample_freq=config.sde_sample_freq,
            target_kl=config.target_kl, penalty_initial_value=config.penalty_initial_value,
            penalty_learning_rate=config.penalty_learning_rate, budget=config.budget, seed=config.seed, device=dev, verbose=0,
            algo_type="pidlagrangian" if getattr(config, "use_pid", False) else "lagrangian",
            pid_kwargs=dict(alpha=config.budget, penalty_init=config.penalty_initial_value, Kp=config.proportional_control_coeff,
                            Ki=config.integral_control_coeff, Kd=config.derivative_control_coeff, pid_delay=config.pid_delay,
                            delta_p_ema_alpha=config.proportional_cost_ema_alpha, delta_d_ema_alpha=config.derivative_cost_ema_alpha),
            policy_kwargs=dict(net_arch=utils.get_net_arch(config)),
            action_noise=getattr(config, "action_noise", "device"), permutation=getattr(config, "permutation", "numpy"),
            streams=getattr(config, "streams", None))
        # the constructor seeded every generator with config.seed (common/utils.py:23-39): all ranks now hold the SAME initial
        # networks; from here on rank r draws its own action noise / minibatch permutations
        D.decorrelate_streams(config.seed, rank)
        return agent

    st = dict(config=config, rank=rank, world=world, train_env=train_env, sampling_env=sampling_env, eval_env=eval_env,
              constraint_net=constraint_net, create_nominal_agent=create_nominal_agent, agent=create_nominal_agent(),
              expert_agent=expert_agent, true_cost_function=get_true_cost_function(config.eval_env_id),
              d_expert_obs=torch.as_tensor(np.asarray(expert_obs), device=dev),
              d_expert_acs=torch.as_tensor(np.asarray(expert_acs), device=dev),
              timesteps=0., start_time=time.time(),
              best=dict(reward=-np.inf, cost=np.inf, fkl=np.inf, rkl=np.inf))
    if world > 1 or getattr(config, "force_collective", False):     # common history of the running moments for the exact cross-rank merge: the (identical) initial state
        st["rms_list"] = [train_env.obs_rms, train_env.ret_rms, train_env.cost_rms]
        st["rms_prev"] = [D.moments_to_sums(r.mean, r.var, r.count) for r in st["rms_list"]]
    if config.warmup_timesteps is not None:    # ref: icrl/icrl.py:185-193 — no cost is incurred during the warm-up
        st["agent"].learn(total_timesteps=config.warmup_timesteps, cost_function=null_cost)
        st["timesteps"] += st["agent"].num_timesteps
        synchronise(st)                        # the ranks' warm-ups saw different shards: merge before the loop starts
    return st


def synchronise(st):
    """the single collective of an outer iteration (no-op on one rank): average policy / constraint-net parameters and Adam
    moments and the dual variable, agree on the step counters, merge the three running-moment sets exactly."""
    force = bool(getattr(st["config"], "force_collective", False))      # bench.py's scale anchor: ONE rank through the multi-rank path
    if st["world"] <= 1 and not force:
        return
    agent, cn = st["agent"], st["constraint_net"]
    pol, dual = agent.policy, agent.dual
    if hasattr(dual, "log_nu"):
        scal = D.Scalars(avg=[(dual, "log_nu"), (dual, "m"), (dual, "v")],
                         counters=[(pol, "adam_step"), (cn, "adam_step"), (dual, "t")])
    else:       # PIDLagrangian: controller state is averaged; its derivative history (a deque of past EMAs) stays per rank
        scal = D.Scalars(avg=[(dual, "pid_i"), (dual, "cost_penalty"), (dual, "_delta_p"), (dual, "_cost_delta")],
                         counters=[(pol, "adam_step"), (cn, "adam_step")])
    ev, t0 = None, time.perf_counter()
    if st.get("sync_events") is not None:      # (bench.py: the collective's share of an outer iteration, device time pack -> unpack)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)); ev[0].record()
    st["rms_prev"] = D.allreduce_state([pol.params, pol.exp_avg, pol.exp_avg_sq, cn.params, cn.exp_avg, cn.exp_avg_sq],
                                       st["rms_list"], st["rms_prev"], st["world"], scalars=scal, force_collective=force)
    pol.prepare(); cn.prepare()
    if ev is not None:
        ev[1].record(); st["sync_events"].append(ev); st["sync_host_ms"].append(1e3 * (time.perf_counter() - t0))


def outer_iteration(st, itr):
    """one pass of the loop body (ref: icrl/icrl.py:199-304)."""
    config, rank, world = st["config"], st["rank"], st["world"]
    train_env, sampling_env, eval_env, constraint_net = st["train_env"], st["sampling_env"], st["eval_env"], st["constraint_net"]
    if config.reset_policy and itr != 0:
        old_agent = st["agent"]
        st["agent"] = st["create_nominal_agent"]()
        if hasattr(old_agent, "_train_ws"):         # same shapes: the update workspace (and where it was found to be fastest) carries over
            st["agent"]._train_ws = old_agent._train_ws
    nominal_agent = st["agent"]
    current_progress_remaining = 1 - float(itr) / float(config.n_iters)
    # ---- forward step
    nominal_agent.learn(total_timesteps=config.forward_timesteps, cost_function="cost")
    forward_metrics = dict(logger.Logger.CURRENT.name_to_value)
    st["timesteps"] += nominal_agent.num_timesteps
    # ---- nominal trajectories
    sync_envs_normalization(train_env, sampling_env)
    streams = getattr(config, "streams", None)       # explicit random streams (tests: teacher forcing); None = device RNG
    A = 1 if nominal_agent.policy.discrete else nominal_agent.policy.act_dim
    orig_observations, observations, actions, rewards, lengths = utils.sample_from_agent(
        nominal_agent, sampling_env, config.expert_rollouts,
        noise=None if streams is None else streams.sample_noise(config.expert_rollouts * sampling_env.unwrapped.max_steps, A))
    # ---- backward step
    mean, var = None, None
    if config.cn_normalize:
        mean, var = sampling_env.obs_rms.mean, sampling_env.obs_rms.var
    cn_perms = None       # minibatch mode with run-private streams: the permutations of get() come from them (seed batches)
    if constraint_net.batch_size is not None and streams is not None and hasattr(streams, "cn_permutations"):
        cn_perms = streams.cn_permutations(config.backward_iters, min(int(orig_observations.shape[0]), int(np.asarray(constraint_net.expert_obs).shape[0])))
    backward_metrics = constraint_net.train(config.backward_iters, orig_observations, actions, lengths, mean, var,
                                            current_progress_remaining, perms=cn_perms)
    train_env.set_cost_function(constraint_net.cost_function)
    # ---- the single collective of the iteration
    synchronise(st)
    # ---- evaluation
    average_true_cost = mean_cost(st["true_cost_function"], orig_observations, actions)
    samples_behind = float((orig_observations[..., 0] < -3).double().mean().item())
    samples_infront = float((orig_observations[..., 0] > 3).double().mean().item())
    sync_envs_normalization(train_env, eval_env)
    average_true_reward, std_true_reward = utils.evaluate_policy(
        nominal_agent, eval_env, n_eval_episodes=10, deterministic=False,
        noise=None if streams is None else streams.eval_noise(10 * eval_env.unwrapped.max_steps, A))
    forward_kl = reverse_kl = float("nan")
    if st["expert_agent"] is not None:
        forward_kl = utils.compute_kl(nominal_agent, st["d_expert_obs"], st["d_expert_acs"], st["expert_agent"])
        reverse_kl = utils.compute_kl(st["expert_agent"], orig_observations, actions, nominal_agent)
    # ---- save (ref: icrl.py:254-269)
    best = st["best"]
    if config.save_dir and itr % config.save_every == 0 and rank == 0:
        path = os.path.join(config.save_dir, f"models/icrl_{itr}_itrs")
        os.makedirs(path, exist_ok=True)
        nominal_agent.save(os.path.join(path, "nominal_agent"))                              # SB3-style archive (ref: icrl.py:259)
        torch.save(nominal_agent.policy.state_dict(), os.path.join(path, "nominal_agent_policy.pth"))
        constraint_net.save(os.path.join(path, "cn.pt"))
        train_env.save(os.path.join(path, f"{itr}_train_env_stats.pkl"))
    if average_true_reward > best["reward"] and config.save_dir and rank == 0:
        nominal_agent.save(os.path.join(config.save_dir, "best_nominal_model"))          # SB3-style archive (ref: icrl.py:266)
        torch.save(nominal_agent.policy.state_dict(), os.path.join(config.save_dir, "best_nominal_model_policy.pth"))
        constraint_net.save(os.path.join(config.save_dir, "best_cn_model.pt"))
        train_env.save(os.path.join(config.save_dir, "train_env_stats.pkl"))
    best["reward"] = max(best["reward"], average_true_reward)
    best["cost"] = min(best["cost"], average_true_cost)
    best["fkl"] = min(best["fkl"], forward_kl) if forward_kl == forward_kl else best["fkl"]
    best["rkl"] = min(best["rkl"], reverse_kl) if reverse_kl == reverse_kl else best["rkl"]
    metrics = {"time(m)": (time.time() - st["start_time"]) / 60, "iteration": itr, "timesteps": st["timesteps"],
               "true/reward": average_true_reward, "true/reward_std": std_true_reward, "true/cost": average_true_cost,
               "true/samples_infront": samples_infront, "true/samples_behind": samples_behind,
               "true/forward_kl": forward_kl, "true/reverse_kl": reverse_kl, "best_true/best_reward": best["reward"],
               "best_true/best_cost": best["cost"], "best_true/best_forward_kl": best["fkl"],
               "best_true/best_reverse_kl": best["rkl"]}
    metrics.update({k.replace("train/", "forward/"): v for k, v in forward_metrics.items()})
    metrics.update(backward_metrics)
    return metrics


def icrl(config, log=print):
    st = setup(config)
    all_metrics = []
    for itr in range(config.n_iters):
        metrics = outer_iteration(st, itr)
        all_metrics.append(metrics)
        if config.verbose > 0 and st["rank"] == 0 and log is not None:
            log(json.dumps({k: (round(float(v), 6) if isinstance(v, (int, float, np.floating, np.integer)) else str(v))
                            for k, v in metrics.items()}))
    return all_metrics, st["agent"], st["constraint_net"], st["train_env"]


def build_parser():
    """flag set of the reference (icrl/icrl.py:316-417); W&B / plotting flags are accepted and ignored."""
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("file_to_run", type=str, nargs="?", default="icrl")
    a("--config_file", "-cf", type=str, default=None); a("--project", "-p", type=str, default="ABC")
    a("--name", "-n", type=str, default=None); a("--group", "-g", type=str, default=None)
    a("--device", "-d", type=str, default="cuda"); a("--verbose", "-v", type=int, default=2)
    a("--sync_wandb", "-sw", action="store_true"); a("--wandb_sweep", "-ws", type=bool, default=False)
    a("--train_env_id", "-tei", type=str, default="HCWithPos-v0"); a("--eval_env_id", "-eei", type=str, default="HCWithPosTest-v0")
    a("--dont_normalize_obs", "-dno", action="store_true"); a("--dont_normalize_reward", "-dnr", action="store_true")
    a("--dont_normalize_cost", "-dnc", action="store_true"); a("--seed", "-s", type=int, default=None)
    a("--clip_obs", "-co", type=int, default=20); a("--cost_info_str", "-cis", type=str, default="cost")
    a("--policy_name", "-pn", type=str, default="TwoCriticsMlpPolicy"); a("--shared_layers", "-sl", type=int, default=None, nargs="*")
    a("--policy_layers", "-pl", type=int, default=[64, 64], nargs="*"); a("--reward_vf_layers", "-rvl", type=int, default=[64, 64], nargs="*")
    a("--cost_vf_layers", "-cvl", type=int, default=[64, 64], nargs="*")
    a("--n_steps", "-ns", type=int, default=2048); a("--batch_size", "-bs", type=int, default=64); a("--n_epochs", "-ne", type=int, default=10)
    a("--num_threads", "-nt", type=int, default=5); a("--save_every", "-se", type=float, default=1); a("--eval_every", "-ee", type=float, default=2048)
    a("--reward_gamma", "-rg", type=float, default=0.99); a("--reward_gae_lambda", "-rgl", type=float, default=0.95)
    a("--cost_gamma", "-cg", type=float, default=0.99); a("--cost_gae_lambda", "-cgl", type=float, default=0.95)
    a("--clip_range", "-cr", type=float, default=0.2); a("--clip_range_reward_vf", "-crv", type=float, default=None)
    a("--clip_range_cost_vf", "-ccv", type=float, default=None); a("--ent_coef", "-ec", type=float, default=0.)
    a("--reward_vf_coef", "-rvc", type=float, default=0.5); a("--cost_vf_coef", "-cvc", type=float, default=0.5)
    a("--target_kl", "-tk", type=float, default=None); a("--max_grad_norm", "-mgn", type=float, default=0.5)
    a("--learning_rate", "-lr", type=float, default=3e-4)
    a("--use_pid", "-upid", action="store_true"); a("--penalty_initial_value", "-piv", type=float, default=1)
    a("--budget", "-b", type=float, default=0.0); a("--update_penalty_after", "-upa", type=int, default=1)
    a("--proportional_control_coeff", "-kp", type=float, default=10); a("--derivative_control_coeff", "-kd", type=float, default=0)
    a("--integral_control_coeff", "-ki", type=float, default=0.0001); a("--proportional_cost_ema_alpha", "-pema", type=float, default=0.5)
    a("--derivative_cost_ema_alpha", "-dema", type=float, default=0.5); a("--pid_delay", "-pidd", type=int, default=1)
    a("--penalty_learning_rate", "-plr", type=float, default=0.1)
    a("--use_sde", "-us", action="store_true"); a("--use_curiosity_driven_exploration", "-ucde", action="store_true")
    a("--sde_sample_freq", "-ssf", type=int, default=-1)
    a("--train_gail_lambda", "-tgl", action="store_true"); a("--n_iters", "-ni", type=int, default=100)
    a("--warmup_timesteps", "-wt", type=lambda x: int(float(x)), default=None)
    a("--forward_timesteps", "-ft", type=lambda x: int(float(x)), default=1e6); a("--backward_iters", "-bi", type=int, default=10)
    a("--no_importance_sampling", "-nis", action="store_true"); a("--per_step_importance_sampling", "-psis", action="store_true")
    a("--reset_policy", "-rp", action="store_true")
    a("--cn_layers", "-cl", type=int, default=[64, 64], nargs="*"); a("--anneal_clr_by_factor", "-aclr", type=float, default=1.0)
    a("--cn_learning_rate", "-clr", type=float, default=3e-4); a("--cn_reg_coeff", "-crc", type=float, default=0)
    a("--cn_batch_size", "-cbs", type=int, default=None); a("--cn_obs_select_dim", "-cosd", type=int, default=None, nargs="+")
    a("--cn_acs_select_dim", "-casd", type=int, default=None, nargs="+"); a("--cn_plot_every", "-cpe", type=int, default=1)
    a("--cn_normalize", "-cn", action="store_true"); a("--cn_target_kl_old_new", "-ctkon", type=float, default=10)
    a("--cn_target_kl_new_old", "-ctkno", type=float, default=10); a("--cn_eps", "-ce", type=float, default=1e-5)
    a("--expert_path", "-ep", type=str, default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/expert_hc.npz"))
    a("--expert_rollouts", "-er", type=int, default=20)
    # additions of this build
    a("--expert_agent_path", type=str, default=None, help="agent zip / npz with policy.pth for the KL metrics")
    a("--save_dir", type=str, default=None); a("--action_noise", type=str, default="device"); a("--permutation", type=str, default="numpy")
    return p


def explicit_dests(parser, argv):
    """dests of the options that appear on the command line under ANY of their spellings (-s / --seed): the reference maps
    short and long names through one table (icrl/utils.py:176-219).  Precedence: command line > config file > default."""
    by_flag = {s: act.dest for act in parser._actions for s in act.option_strings}
    return {by_flag[a.split("=")[0]] for a in argv if a.split("=")[0] in by_flag}


def main(argv=None):
    start = time.time()
    parser = build_parser()
    args = parser.parse_args(argv if argv is not None else sys.argv[1:])
    config = vars(args)
    if config["config_file"] is not None and config["config_file"].endswith(".json"):
        with open(config["config_file"]) as f:
            file_cfg = json.load(f)
        config.update({k: v for k, v in file_cfg.items() if k not in explicit_dests(parser, argv if argv is not None else sys.argv[1:])})
    rank, world = D.init_from_env()
    if config["seed"] is None and rank == 0:
        config["seed"] = int(np.random.randint(0, 100))
    config["seed"] = D.broadcast_seed(config["seed"], rank, world)     # every rank builds the same initial networks
    config["rank"], config["world_size"] = rank, world
    if config["save_dir"]:
        os.makedirs(config["save_dir"], exist_ok=True)
        with open(os.path.join(config["save_dir"], "config.json"), "w") as f:
            json.dump({k: v for k, v in config.items()}, f, indent=2, default=str)
    icrl(types.SimpleNamespace(**config))
    if rank == 0:
        print("Time taken: %05.2f hours" % ((time.time() - start) / 3600))


if __name__ == "__main__":
    main()
