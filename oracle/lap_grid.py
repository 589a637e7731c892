"""Oracle: LapGridWorld / ConstrainedLapGridWorld, vectorised.  Test infrastructure only.

Exact restatement of the reference's only dependency-free environment (BASELINE configs[0]):

  ref: custom_envs/custom_envs/envs/lap_grid_world.py:29-119   (LapGridWorld: 40 cells on a square lap, coins worth 3 at
       cells 5 / 15 / 25 / 35, action 0 = forward, 1 = backward, wrap-around, 200-step episodes, no randomness)
       lap_grid_world.py:197-203                                (obs = ((pos - 0) * 2) / 40 - 1 in float64)
       lap_grid_world.py:205-240                                (Constrained variant: backward = reward -1 and done, position kept)
       custom_envs/custom_envs/__init__.py:357-370              (LGW-v0 / CLGW-v0, max_episode_steps 200)
       stable_baselines3/common/vec_env/dummy_vec_env.py:43-58  (auto-reset: the returned observation is the reset one)

Same interface as oracle/synth_env.py:SynthVecEnv so that EnvStack / PortAgent step it unchanged.
"""
import numpy as np

F64 = np.float64
N_CELLS = 40
EPISODE = 200
COINS = (5, 15, 25, 35)


def cell_rewards():
    r = np.zeros(N_CELLS, F64)
    r[list(COINS)] = 3.0
    return r


def obs_of(pos):
    """normalize_obs on an int64 position: (pos - 0.0) * 2 / 40.0 - 1, one float64 rounding per operation."""
    o = np.asarray(pos).astype(F64)
    o = o * 2.0
    o = o / 40.0
    return o - 1.0


class LapGridVecEnv:
    kind = "lgw"
    obs_dim, act_dim, max_steps = 1, 2, EPISODE
    discrete = True
    action_low = action_high = None

    def __init__(self, n_envs, constrained=False, seed=0):
        self.n_envs = n_envs
        self.constrained = constrained
        self.rewards = cell_rewards()
        self.seed(seed)

    def seed(self, seed=0, env_index_offset=0):
        self.pos = np.zeros(self.n_envs, np.int64)
        self.t_ep = np.zeros(self.n_envs, np.int64)
        self.s = obs_of(self.pos)[:, None]

    def reset(self):
        self.pos[:] = 0
        self.t_ep[:] = 0
        self.s = obs_of(self.pos)[:, None]
        return self.s.copy()

    def step(self, actions):
        a = np.asarray(actions).reshape(self.n_envs).astype(np.int64)
        fwd, back = a == 0, a == 1
        done = np.zeros(self.n_envs, bool)
        pos = self.pos.copy()
        pos[fwd] = (pos[fwd] + 1) % N_CELLS
        if self.constrained:
            rew = np.where(fwd, self.rewards[pos], -1.0)
            done |= back
        else:
            pos[back] = (pos[back] - 1) % N_CELLS
            rew = self.rewards[pos].copy()
        self.t_ep += 1
        done |= self.t_ep >= self.max_steps
        pos[done] = 0
        self.t_ep[done] = 0
        self.pos = pos
        self.s = obs_of(pos)[:, None]
        return self.s.copy(), rew.astype(F64), done
