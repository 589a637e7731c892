"""Ground-truth cost functions used for the `true/cost` metric (ref: icrl/true_constraint_net.py:11-55)."""
from functools import partial

import numpy as np
import torch


def wall_behind(pos, obs, acs):
    return obs[..., 0] <= pos


def null_cost(x, *args):
    return np.zeros(x.shape[:1])


def lap_grid_world(obs, acs):
    """ref: icrl/true_constraint_net.py:104-111 — the backward action (index 1) is the constraint violation."""
    a = acs.reshape(acs.shape[0], -1)[:, 0]
    return a == 1


def get_true_cost_function(env_id):
    if env_id == "CLGW-v0":
        return lap_grid_world
    if env_id in ("HCWithPosTest-v0", "WalkerWithPosTest-v0", "SwimmerWithPosTest-v0", "AntWallTest-v0", "AntWallBrokenTest-v0"):
        return partial(wall_behind, -3)
    return null_cost


def mean_cost(fn, obs, acs):
    c = fn(obs, acs)
    return float(c.double().mean().item()) if torch.is_tensor(c) else float(np.mean(c))
