"""Expert-rollout writer in the reference's on-disk format (ref: icrl/run_policy.py:82-103, icrl/utils.py save_dict_as_pkl):

    <save_dir>/rollouts/{i}.pkl = dict(observations float64 [L, obs] (un-normalised, the observation AFTER each step),
                                       actions float32 [L, act], rewards float64 [1], lengths int64 [1], save_scheme='not_airl')

so that rollouts sampled with this build can be consumed by the reference (`icrl.py -ep <dir>`) and vice versa
(`utils.load_expert_data` reads either).  One episode per file, optional reward / length thresholds like the reference.
"""
import os
import pickle
import shutil

import numpy as np

from . import utils


def save_rollouts(model, env, n_rollouts, save_dir, reward_threshold=None, length_threshold=None, max_tries=None):
    """`env`: a 1-env evaluation stack (utils.make_eval_env).  Returns the list of written paths."""
    rollouts_dir = os.path.join(save_dir, "rollouts")
    shutil.rmtree(rollouts_dir, ignore_errors=True)                    # del_and_make (icrl/utils.py)
    os.makedirs(rollouts_dir)
    paths, tries = [], 0
    while len(paths) < n_rollouts and (max_tries is None or tries < max_tries):
        tries += 1
        orig_obs, _, actions, rewards, lengths = utils.sample_from_agent(model, env, 1)
        d = dict(observations=orig_obs.cpu().numpy().astype(np.float64), actions=actions.cpu().numpy().astype(np.float32),
                 rewards=np.asarray(rewards, np.float64), lengths=np.asarray(lengths, np.int64), save_scheme="not_airl")
        if model.policy.discrete:
            d["actions"] = d["actions"].reshape(-1)                    # np.squeeze(np.array(actions), axis=1) of int actions
        if (reward_threshold is None or np.mean(d["rewards"]) >= reward_threshold) and \
           (length_threshold is None or np.mean(d["lengths"]) >= length_threshold):
            path = os.path.join(rollouts_dir, f"{len(paths)}.pkl")
            with open(path, "wb") as f:
                pickle.dump(d, f)
            paths.append(path)
    return paths


def run_policy(args):
    """ref: icrl/run_policy.py:20-103 — load a trained agent from a run directory and write `n_rollouts` episodes.

    The run directory is what `run_me.py icrl / cpg --save_dir` writes (config.json, best_nominal_model.zip | best_model.zip,
    models/..., train_env_stats.pkl), or a reference W&B run directory (the same files under `<load_dir>/files`).  Videos
    (`eval_and_make_video`) and the AIRL saving scheme are outside the hot path and not provided; `--remote` (W&B restore) needs
    network access."""
    import json
    import types
    from .ppo_lag import PPOLagrangian
    from .vec_env import VecNormalize
    if args.remote:
        raise NotImplementedError("run_policy --remote restores files from the W&B server; copy the run directory instead")
    if args.save_using_airl_scheme:
        raise NotImplementedError("run_policy --save_using_airl_scheme: the AIRL baseline is not part of this build")
    if args.is_icrl:
        f = f"models/icrl_{args.load_itr}_itrs/nominal_agent" if args.load_itr is not None else "best_nominal_model"
    else:
        f = f"models/rl_model_{args.load_itr}_steps" if args.load_itr is not None else "best_model"
    load_dir = os.path.join(args.load_dir, "files") if os.path.isdir(os.path.join(args.load_dir, "files")) else args.load_dir
    with open(os.path.join(load_dir, "config.json")) as fh:
        config = types.SimpleNamespace(**json.load(fh))
    save_dir = os.path.join(load_dir, args.save_dir)
    # the directory is deleted first (del_and_make, icrl/utils.py): only ever a proper sub-directory of the run directory
    real_load, real_save = os.path.realpath(load_dir), os.path.realpath(save_dir)
    if real_save == real_load or not real_save.startswith(real_load + os.sep):
        raise ValueError(f"--save_dir {args.save_dir!r} must name a sub-directory of the run directory {load_dir!r}")
    shutil.rmtree(save_dir, ignore_errors=True)
    os.makedirs(save_dir)
    env_id = args.env_id or config.eval_env_id
    env = utils.make_eval_env(env_id, use_cost_wrapper=False, normalize_obs=False)
    if not getattr(config, "dont_normalize_obs", False):
        env = VecNormalize.load(os.path.join(load_dir, "train_env_stats.pkl"), env)        # restore the training statistics
        env.norm_reward = False
        env.training = False
    model = PPOLagrangian.load(os.path.join(load_dir, f), env=env)
    if args.dont_save_trajs:
        return []
    paths = save_rollouts(model, env, args.n_rollouts, save_dir, args.reward_threshold, args.length_threshold)
    for i, pth in enumerate(paths):
        with open(pth, "rb") as fh:
            d = pickle.load(fh)
        print(f"{i}. Mean reward: {np.mean(d['rewards'])} | Mean length: {np.mean(d['lengths'])}")
    return paths


def build_parser():
    """flag set of the reference (icrl/run_policy.py:105-119)."""
    import argparse
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("file_to_run", type=str, nargs="?", default="run_policy")
    a("--load_dir", "-l", type=str, default="icrl/wandb/latest-run/"); a("--is_icrl", "-ii", action="store_true")
    a("--remote", "-r", action="store_true"); a("--save_dir", "-s", type=str, default="run_policy")
    a("--env_id", "-e", type=str, default=None); a("--load_itr", "-li", type=int, default=None)
    a("--n_rollouts", "-nr", type=int, default=3); a("--dont_make_video", "-dmv", action="store_true")
    a("--dont_save_trajs", "-dst", action="store_true"); a("--save_using_airl_scheme", "-suas", action="store_true")
    a("--reward_threshold", "-rt", type=float, default=None); a("--length_threshold", "-lt", type=int, default=None)
    return p


def main(argv=None):
    return run_policy(build_parser().parse_args(argv))
