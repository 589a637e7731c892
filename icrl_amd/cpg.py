"""cpg — PPO-Lagrangian against a FIXED cost: constraint transfer (BASELINE configs[4]) and expert generation.

ref: icrl/cpg.py:24-212 (cpg), :214-343 (flag set).  The cost is the frozen ConstraintNet loaded through the reference's
(positionally shifted) ConstraintNet.load, the ground-truth wall cost, or the null cost.  One long
``model.learn(timesteps, cost_function="cost")`` with the reference's callbacks (icrl/cpg.py:160-198): CheckpointCallback every
``save_every`` calls, EvalCallback every ``eval_every`` calls (5 stochastic episodes on the cost-wrapped 1-env test stack,
best model + env statistics saved), AdjustedRewardCallback after every rollout — icrl_amd/callbacks.py, counted in
vectorised env steps like the reference.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

from . import callbacks, distributed as D, logger, utils
from .constraint_net import ConstraintNet
from .ppo_lag import PPOLagrangian
from .true_constraint_net import get_true_cost_function, null_cost
from . import spaces


class _History(callbacks.BaseCallback):
    """one record per rollout of what the other callbacks logged (returned by cpg() for the tests / the caller)."""

    def __init__(self, log):
        super().__init__()
        self.log, self.history = log, []

    def _on_rollout_end(self):
        lg = logger.Logger.CURRENT.name_to_value
        rec = {"rollouts": len(self.history) + 1, "timesteps": self.model.num_timesteps, "nu": float(self.model.dual.nu().item())}
        for k in ("rollout/adjusted_reward", "eval/true_cost", "eval/mean_reward", "eval/best_mean_reward"):
            if k in lg:
                rec[k] = float(lg[k])
        self.history.append(rec)
        if self.log:
            self.log(json.dumps(rec))


def setup(config, log=print):
    """everything of cpg() up to the learn() call: (model, callback, cost argument of learn(), history callback).  bench.py times
    learn() calls on the result; cpg() below is setup + one learn()."""
    logger.configure()          # a fresh scalar log per run (the reference configures its logger in learn())
    rank, world = getattr(config, "rank", 0), getattr(config, "world_size", 1)
    dev = config.device if str(config.device).startswith("cuda") else "cuda"
    train_env = utils.make_train_env(env_id=config.train_env_id, save_dir=config.save_dir, use_cost_wrapper=True,
                                     base_seed=config.seed, num_threads=config.num_threads,
                                     normalize_obs=not config.dont_normalize_obs, normalize_reward=not config.dont_normalize_reward,
                                     normalize_cost=not config.dont_normalize_cost, cost_info_str=config.cost_info_str,
                                     reward_gamma=config.reward_gamma, cost_gamma=config.cost_gamma,
                                     env_index_offset=rank * config.num_threads, device=dev)
    eval_env = utils.make_eval_env(env_id=config.eval_env_id, use_cost_wrapper=True, normalize_obs=not config.dont_normalize_obs,
                                   seed=config.seed + rank * config.num_threads, device=dev)
    is_discrete = isinstance(train_env.action_space, spaces.Discrete)
    obs_dim = train_env.observation_space.shape[0]
    acs_dim = train_env.action_space.n if is_discrete else train_env.action_space.shape[0]
    if config.use_null_cost:
        cost_function = null_cost
    elif config.cn_path is None:
        cost_function = get_true_cost_function(config.eval_env_id)
    elif config.load_gail:
        # ref: icrl/cpg.py:54-83 — the cost is the discriminator's output D itself (apply_log=False); numpy in / numpy out like the
        # reference's callable, so it runs through VecCostWrapper's callable branch and the per-step rollout loop
        from .gail_utils import GailDiscriminator
        action_low, action_high = None, None
        if isinstance(train_env.action_space, spaces.Box):
            action_low, action_high = train_env.action_space.low, train_env.action_space.high
        gail = GailDiscriminator.load(config.cn_path, obs_dim=obs_dim, acs_dim=acs_dim, is_discrete=is_discrete,
                                      obs_select_dim=config.cn_obs_select_dim, acs_select_dim=config.cn_acs_select_dim,
                                      clip_obs=None, obs_mean=None, obs_var=None, action_low=action_low, action_high=action_high)

        def cost_function(obs, acs):
            return gail.reward_function(obs, acs, apply_log=False).cpu().numpy()
    else:
        constraint_net = ConstraintNet.load(config.cn_path, obs_dim=obs_dim, acs_dim=acs_dim, is_discrete=is_discrete,
                                            obs_select_dim=config.cn_obs_select_dim, acs_select_dim=config.cn_acs_select_dim,
                                            clip_obs=None, obs_mean=None, obs_var=None)
        cost_function = constraint_net.cost_function
    # a ConstraintNet cost runs inside the fused rollout launch; null_cost / the analytic true cost (numpy callables) go
    # through VecCostWrapper's callable branch and the per-step rollout loop (PPOLagrangian._collect_rollouts_stepped)
    train_env.set_cost_function(cost_function)
    eval_env.set_cost_function(cost_function)
    model = PPOLagrangian(
        policy=config.policy_name, env=train_env, algo_type="pidlagrangian" if config.use_pid else "lagrangian",
        learning_rate=config.learning_rate, n_steps=config.n_steps, batch_size=config.batch_size, n_epochs=config.n_epochs,
        reward_gamma=config.reward_gamma, reward_gae_lambda=config.reward_gae_lambda, cost_gamma=config.cost_gamma,
        cost_gae_lambda=config.cost_gae_lambda, clip_range=config.clip_range, clip_range_reward_vf=config.clip_range_reward_vf,
        clip_range_cost_vf=config.clip_range_cost_vf, ent_coef=config.ent_coef, reward_vf_coef=config.reward_vf_coef,
        cost_vf_coef=config.cost_vf_coef, max_grad_norm=config.max_grad_norm, use_sde=config.use_sde,
        sde_sample_freq=config.sde_sample_freq, target_kl=config.target_kl, penalty_initial_value=config.penalty_initial_value,
        penalty_learning_rate=config.penalty_learning_rate, update_penalty_after=config.update_penalty_after, budget=config.budget,
        seed=config.seed, device=dev, verbose=config.verbose,
        pid_kwargs=dict(alpha=config.budget, penalty_init=config.penalty_initial_value, Kp=config.proportional_control_coeff,
                        Ki=config.integral_control_coeff, Kd=config.derivative_control_coeff, pid_delay=config.pid_delay,
                        delta_p_ema_alpha=config.proportional_cost_ema_alpha, delta_d_ema_alpha=config.derivative_cost_ema_alpha),
        policy_kwargs=dict(net_arch=utils.get_net_arch(config)),
        action_noise=getattr(config, "action_noise", "device"), permutation=getattr(config, "permutation", "numpy"),
        streams=getattr(config, "streams", None))
    if world > 1:      # all ranks hold the same initial networks (same seed); from here on rank r draws its own noise / permutations
        D.decorrelate_streams(config.seed, rank)
    # ref: icrl/cpg.py:160-176
    hist = _History(log if (config.verbose > 0 and rank == 0) else None)
    eval_every = int(config.eval_every) if not getattr(config, "eval_every_rollouts", 0) else int(config.eval_every_rollouts) * int(config.n_steps)
    cbs = [callbacks.EvalCallback(eval_env, eval_freq=eval_every, best_model_save_path=config.save_dir if rank == 0 else None,
                                  deterministic=False, verbose=0,
                                  callback_on_new_best=callbacks.SaveEnvStatsCallback(train_env, config.save_dir if rank == 0 else None)),
           callbacks.AdjustedRewardCallback(get_true_cost_function(config.eval_env_id)), hist]
    if config.save_dir and rank == 0:
        cbs.insert(0, callbacks.CheckpointCallback(int(config.save_every), os.path.join(config.save_dir, "models"), verbose=0))
    if world > 1 or getattr(config, "force_collective", False):      # env shards (BASELINE configs[4]: 4096 envs over 8 GPUs): one all-reduce per rollout + update
        cbs.insert(0, callbacks.RankSyncCallback(train_env, world, force_collective=bool(getattr(config, "force_collective", False))))
    cb = callbacks.CallbackList(cbs)
    # ref: icrl/cpg.py:201-203 — `-cis None` hands the callable itself to learn() (costs evaluated outside the env chain)
    learn_cost = config.cost_info_str if config.cost_info_str is not None else cost_function
    return model, cb, learn_cost, hist


def cpg(config, log=print):
    rank = getattr(config, "rank", 0)
    model, cb, learn_cost, hist = setup(config, log)
    model.learn(total_timesteps=int(config.timesteps), cost_function=learn_cost, callback=cb)
    if config.save_dir and rank == 0:
        torch.save(model.policy.state_dict(), os.path.join(config.save_dir, "final_model_policy.pth"))
    return model, hist.history


def build_parser():
    """flag set of the reference's cpg (icrl/cpg.py:216-298); note -cl here is --cost_vf_layers."""
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("file_to_run", type=str, nargs="?", default="cpg")
    a("--config_file", "-cf", type=str, default=None); a("--project", "-p", type=str, default="ABC"); a("--name", "-n", type=str, default=None)
    a("--group", "-g", type=str, default=None); a("--message", "-m", type=str, default=None); a("--device", "-d", type=str, default="cuda")
    a("--verbose", "-v", type=int, default=2); a("--wandb_sweep", "-ws", type=bool, default=False); a("--sync_wandb", "-sw", action="store_true")
    a("--cost_info_str", "-cis", type=lambda x: None if str(x).lower() == "none" else str(x), default="cost")
    a("--train_env_id", "-tei", type=str, default="AntWallBroken-v0"); a("--eval_env_id", "-eei", type=str, default="AntWallBrokenTest-v0")
    a("--dont_normalize_obs", "-dno", action="store_true"); a("--dont_normalize_reward", "-dnr", action="store_true")
    a("--dont_normalize_cost", "-dnc", action="store_true"); a("--seed", "-s", type=int, default=None)
    a("--policy_name", "-pn", type=str, default="TwoCriticsMlpPolicy"); a("--shared_layers", "-sl", type=int, default=None, nargs="*")
    a("--policy_layers", "-pl", type=int, default=[64, 64], nargs="*"); a("--reward_vf_layers", "-rl", type=int, default=[64, 64], nargs="*")
    a("--cost_vf_layers", "-cl", type=int, default=[64, 64], nargs="*"); a("--cnn_features_dim", "-cfd", type=int, default=512)
    a("--timesteps", "-t", type=lambda x: int(float(x)), default=1e6); a("--n_steps", "-ns", type=int, default=2048)
    a("--batch_size", "-bs", type=int, default=64); a("--n_epochs", "-ne", type=int, default=10); a("--num_threads", "-nt", type=int, default=5)
    a("--save_every", "-se", type=float, default=5e5); a("--eval_every", "-ee", type=float, default=2048); a("--plot_every", "-pe", type=float, default=2048)
    a("--reward_gamma", "-rg", type=float, default=0.99); a("--reward_gae_lambda", "-rgl", type=float, default=0.95)
    a("--cost_gamma", "-cg", type=float, default=0.99); a("--cost_gae_lambda", "-cgl", type=float, default=0.95)
    a("--clip_range", "-cr", type=float, default=0.2); a("--clip_range_reward_vf", "-crv", type=float, default=None)
    a("--clip_range_cost_vf", "-ccv", type=float, default=None); a("--ent_coef", "-ec", type=float, default=0.)
    a("--reward_vf_coef", "-rvc", type=float, default=0.5); a("--cost_vf_coef", "-cvc", type=float, default=0.5)
    a("--target_kl", "-tk", type=float, default=None); a("--max_grad_norm", "-mgn", type=float, default=0.5); a("--learning_rate", "-lr", type=float, default=3e-4)
    a("--use_pid", "-upid", action="store_true"); a("--penalty_initial_value", "-piv", type=float, default=1); a("--budget", "-b", type=float, default=0.0)
    a("--update_penalty_after", "-upa", type=int, default=1); a("--proportional_control_coeff", "-kp", type=float, default=10)
    a("--derivative_control_coeff", "-kd", type=float, default=0); a("--integral_control_coeff", "-ki", type=float, default=0.0001)
    a("--proportional_cost_ema_alpha", "-pema", type=float, default=0.5); a("--derivative_cost_ema_alpha", "-dema", type=float, default=0.5)
    a("--pid_delay", "-pidd", type=int, default=1); a("--penalty_learning_rate", "-plr", type=float, default=0.1)
    a("--use_sde", "-us", action="store_true"); a("--use_curiosity_driven_exploration", "-ucde", action="store_true")
    a("--use_lambda_shaping", "-uls", action="store_true"); a("--sde_sample_freq", "-ssf", type=int, default=-1)
    a("--use_null_cost", "-unc", action="store_true"); a("--cn_path", "-cp", type=str, default=None)
    a("--cn_obs_select_dim", "-cosd", type=int, default=None, nargs="+"); a("--cn_acs_select_dim", "-casd", type=int, default=None, nargs="+")
    a("--cn_device", "-cd", type=str, default=None); a("--load_gail", "-lg", action="store_true")
    a("--save_dir", type=str, default=None); a("--eval_every_rollouts", type=int, default=0)
    a("--action_noise", type=str, default="device"); a("--permutation", type=str, default="numpy")
    return p


def main(argv=None):
    start = time.time()
    config = vars(build_parser().parse_args(argv if argv is not None else sys.argv[1:]))
    if config["seed"] is None:
        config["seed"] = int(np.random.randint(0, 100))
    rank, world = D.init_from_env()
    config["seed"] = D.broadcast_seed(config["seed"], rank, world)     # every rank builds the same initial networks
    config["rank"], config["world_size"] = rank, world
    if config["save_dir"]:
        os.makedirs(config["save_dir"], exist_ok=True)
        with open(os.path.join(config["save_dir"], "config.json"), "w") as fh:      # what run_policy reads back (W&B keeps it in the reference)
            json.dump({k: v for k, v in config.items()}, fh, indent=2, default=str)
    cpg(types.SimpleNamespace(**config))
    if rank == 0:
        print("Time taken: %05.2f hours" % ((time.time() - start) / 3600))


if __name__ == "__main__":
    main()
