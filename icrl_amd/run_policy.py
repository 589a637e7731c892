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
