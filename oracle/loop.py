"""Oracle / CPU port: rollout collection, learn(), nominal sampling, evaluation and the ICRL outer
loop on the synthetic environment.  Test infrastructure only (also timed by bench.py's
``cpu_baseline`` leg as the "port" of the reference's CPU path — same structure: a Python loop
per env step, one optimiser step per minibatch).

ref: stable_baselines3/common/on_policy_algorithm.py:340-421   (collect_rollouts)
     stable_baselines3/common/on_policy_algorithm.py:430-492   (learn)
     stable_baselines3/common/base_class.py:303-366            (_setup_learn: reset on every learn())
     stable_baselines3/common/vec_env/vec_cost_wrapper.py:51-77 (cost from previous raw obs + current action)
     stable_baselines3/ppo_lag/ppo_lag.py:177-338              (train)
     icrl/utils.py:323-357                                      (sample_from_agent: (s_{t+1}, a_t) pairs)
     stable_baselines3/common/evaluation.py:10-67               (evaluate_policy)
     icrl/utils.py:421-437                                      (compute_kl)
     icrl/icrl.py:199-304                                       (outer loop)
"""
import time

import numpy as np
import torch as th

from . import stats
from .cn import cn_train
from .gae import dual_gae
from .nets import CostNet, TwoCriticPolicy
from .ppo import Dual, PID, ppo_lag_train
from .lap_grid import LapGridVecEnv
from .synth_env import SynthVecEnv

F32, F64 = np.float32, np.float64


class EnvStack:
    """SynthVecEnv -> cost wrapper -> VecNormalizeWithCost, as one object with explicit state."""

    def __init__(self, env, norm, cost_fn=None):
        self.env, self.norm, self.cost_fn = env, norm, cost_fn
        self.previous_obs = None
        self.old_obs = self.old_rew = self.old_cost = None

    @property
    def num_envs(self):
        return self.env.n_envs

    def reset(self):
        raw = self.env.reset()
        self.previous_obs = raw                      # ref: vec_cost_wrapper.py:70-77
        self.old_obs = raw
        return stats.norm_reset(self.norm, raw)

    def step(self, clipped_actions):
        raw_obs, raw_rew, dones = self.env.step(clipped_actions)
        cost = None
        if self.cost_fn is not None:                 # ref: vec_cost_wrapper.py:58-65
            cost = self.cost_fn(self.previous_obs.copy(), np.asarray(clipped_actions).copy())
            self.previous_obs = raw_obs.copy()
        self.old_obs, self.old_rew, self.old_cost = raw_obs, raw_rew, cost
        obs_n, rew_n, cost_n = stats.norm_step(self.norm, raw_obs, raw_rew, cost, dones)
        return obs_n, rew_n, dones, cost_n


class Rollout:
    """The 16 [T, N, ...] float32 arrays of RolloutBufferWithCost (ref: buffers.py:468-491)."""

    def __init__(self, T, N, obs_dim, act_dim):
        z = lambda *s: np.zeros((T, N) + s, F32)
        self.T, self.N = T, N
        self.observations, self.new_observations = z(obs_dim), z(obs_dim)
        self.orig_observations, self.new_orig_observations = z(obs_dim), z(obs_dim)
        self.actions = z(act_dim)
        for k in ("dones", "log_probs", "rewards", "reward_returns", "reward_values", "reward_advantages",
                  "costs", "orig_costs", "cost_returns", "cost_values", "cost_advantages"):
            setattr(self, k, z())

    def as_dict(self):
        return {k: v for k, v in self.__dict__.items() if isinstance(v, np.ndarray)}


def explained_variance(y_pred, y_true):
    """ref: stable_baselines3/common/utils.py:43-59."""
    var_y = np.var(y_true)
    return float(np.nan if var_y == 0 else 1 - np.var(y_true - y_pred) / var_y)


class PortAgent:
    """PPO-Lagrangian agent of the CPU port."""

    def __init__(self, stack, *, n_steps=2048, batch_size=64, n_epochs=10, learning_rate=3e-4,
                 reward_gamma=0.99, reward_gae_lambda=0.95, cost_gamma=0.99, cost_gae_lambda=0.95,
                 clip_range=0.2, ent_coef=0.0, reward_vf_coef=0.5, cost_vf_coef=0.5, max_grad_norm=0.5,
                 target_kl=None, penalty_initial_value=1.0, penalty_learning_rate=0.1, budget=0.0,
                 hidden=(64, 64), seed=0, discrete=False, pid_kwargs=None, shared=()):
        self.stack = stack
        env = stack.env
        self.obs_dim, self.act_dim = env.obs_dim, env.act_dim
        self.h = dict(n_steps=n_steps, batch_size=batch_size, n_epochs=n_epochs, clip_range=clip_range,
                      ent_coef=ent_coef, reward_vf_coef=reward_vf_coef, cost_vf_coef=cost_vf_coef,
                      max_grad_norm=max_grad_norm, target_kl=target_kl)
        self.gammas = (reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda)
        # ref: on_policy_algorithm.py:313-316 / common/utils.py:23-39 — seed python/numpy/torch, then build
        import random
        random.seed(seed); np.random.seed(seed); th.manual_seed(seed)
        self.policy = TwoCriticPolicy(self.obs_dim, self.act_dim, hidden, discrete=discrete, shared=shared)
        self.optimizer = th.optim.Adam(self.policy.parameters(), lr=learning_rate, eps=1e-5)  # ref: policies.py:357-361
        # ref: ppo_lag.py:147-160 — algo_type "lagrangian" | "pidlagrangian"
        self.dual = PID(**pid_kwargs) if pid_kwargs is not None else Dual(budget, penalty_learning_rate, penalty_initial_value, None)
        self.discrete = discrete
        self.num_timesteps = 0
        self._n_updates = 0
        self._last_obs = None
        self.buf = None
        self.logs = {}

    # -- rollout -------------------------------------------------------------------------------
    def collect_rollouts(self, noise=None):
        T, N = self.h["n_steps"], self.stack.num_envs
        buf = Rollout(T, N, self.obs_dim, self.act_dim if not self.discrete else 1)
        env = self.stack.env
        for t in range(T):
            with th.no_grad():
                obs_t = th.as_tensor(self._last_obs)
                actions, v_r, v_c, logp = self.policy.forward(obs_t, None if noise is None else th.as_tensor(noise[t]))
            actions = actions.numpy()
            clipped = actions if self.discrete else np.clip(actions, env.action_low, env.action_high)
            new_obs, rewards, dones, costs = self.stack.step(clipped)
            orig_obs = self.stack.old_obs.copy()
            if costs is None:                        # no cost wrapper in the env chain (the GAIL baseline): zero costs
                costs = np.zeros(N, F32)
            orig_costs = costs.copy() if self.stack.old_cost is None else self.stack.old_cost.copy()
            self.num_timesteps += N
            buf.observations[t] = self._last_obs
            buf.orig_observations[t] = self._last_original_obs
            buf.new_observations[t] = new_obs
            buf.new_orig_observations[t] = orig_obs
            buf.actions[t] = actions.reshape(N, -1)
            buf.dones[t] = self._last_dones
            buf.log_probs[t] = logp.numpy()
            buf.rewards[t] = rewards
            buf.reward_values[t] = v_r.numpy().flatten()
            buf.costs[t] = costs
            buf.orig_costs[t] = orig_costs
            buf.cost_values[t] = v_c.numpy().flatten()
            self._last_obs, self._last_original_obs, self._last_dones = new_obs, orig_obs, dones
        # ref: on_policy_algorithm.py:417 — bootstrap values are those of the LAST forward, i.e. of the
        # observation before the final step (reference behaviour, kept).
        g = dual_gae(buf.rewards, buf.costs, buf.reward_values, buf.cost_values, buf.dones,
                     v_r.numpy().flatten(), v_c.numpy().flatten(), dones, *self.gammas)
        buf.reward_returns, buf.reward_advantages = g["reward_returns"], g["reward_advantages"]
        buf.cost_returns, buf.cost_advantages = g["cost_returns"], g["cost_advantages"]
        self.buf = buf
        self.last_values, self.last_dones_out = v_r.numpy().flatten(), dones      # PPO's `extras` (on_policy_algorithm.py:190)
        return buf

    # -- update --------------------------------------------------------------------------------
    def train(self, perms=None):
        n = self.buf.T * self.buf.N
        if perms is None:
            perms = lambda epoch: np.random.permutation(n)          # ref: buffers.py:596
        nu = self.dual.nu().item()
        out = ppo_lag_train(self.policy, self.optimizer, self.buf.as_dict(), perms, nu, discrete=self.discrete, **{
            k: self.h[k] for k in ("batch_size", "n_epochs", "clip_range", "target_kl", "max_grad_norm",
                                   "ent_coef", "reward_vf_coef", "cost_vf_coef")})
        self._n_updates += self.h["n_epochs"]
        average_cost = np.mean(self.buf.orig_costs)
        self.dual.update(average_cost)
        em = lambda x: np.ascontiguousarray(x.T).reshape(-1)
        out.update({"train/nu": self.dual.nu().item(), "train/nu_loss": self.dual.loss.item(),
                    "train/average_cost": float(average_cost), "train/total_cost": float(np.sum(self.buf.orig_costs)),
                    # (after get() the reference's arrays are flattened ENV-major, buffers.py:53-65,600-606: numpy's pairwise sums run in that order)
                    "train/mean_reward_advantages": float(np.mean(em(self.buf.reward_advantages))),
                    "train/mean_cost_advantages": float(np.mean(em(self.buf.cost_advantages))),
                    # ref: ppo_lag.py:311-312 — explained_variance(returns, values): y_pred = returns, y_true = values (argument order kept)
                    "train/reward_explained_variance": explained_variance(em(self.buf.reward_returns), em(self.buf.reward_values)),
                    "train/cost_explained_variance": explained_variance(em(self.buf.cost_returns), em(self.buf.cost_values)),
                    "train/n_updates": self._n_updates,
                    # ref: base_class.py:221 (_update_learning_rate) and ppo_lag.py:333 — constant schedules in this fork's runs
                    "train/learning_rate": float(self.optimizer.param_groups[0]["lr"]), "train/clip_range": float(self.h["clip_range"])})
        if not self.discrete:
            out["train/std"] = th.exp(self.policy.params["log_std"]).mean().item()
        self.logs = out
        return out

    def learn(self, total_timesteps, noise_fn=None, perms_fn=None, streams=None):
        """ref: on_policy_algorithm.py:430-492 with reset_num_timesteps=True: env reset on every call.
        streams: an oracle.streams object; its permutation cursor advances by the EXECUTED epochs of each train()."""
        self.num_timesteps = 0
        self._last_obs = self.stack.reset()
        self._last_dones = np.zeros(self.stack.num_envs, bool)
        self._last_original_obs = self.stack.old_obs.copy()
        T, N, n_epochs = self.h["n_steps"], self.stack.num_envs, self.h["n_epochs"]
        A = 1 if self.discrete else self.act_dim
        it, t_start = 0, time.time()
        while self.num_timesteps < total_timesteps:
            if streams is not None:
                self.collect_rollouts(streams.rollout_noise(T, N, A))
                out = self.train(lambda e: streams.permutation(e, T * N))
                streams.consumed(min(int(out["train/early_stop_epoch"]) + 1, n_epochs))
            else:
                self.collect_rollouts(None if noise_fn is None else noise_fn(it))
                self.train(None if perms_fn is None else perms_fn(it))
            it += 1
        # ref: on_policy_algorithm.py:452-457,488 — the closing training_infos(iteration + 1); wall-clock keys, never compared
        el = max(time.time() - t_start, 1e-9)
        self.logs.update({"time/iterations": it + 1, "time/fps": int(self.num_timesteps / el), "time/time_elapsed": int(el),
                          "time/total_timesteps": self.num_timesteps})
        return self

    # -- inference -----------------------------------------------------------------------------
    def predict(self, obs, noise=None, deterministic=False):
        """ref: policies.py:215-280 — sample, then clip to the action box."""
        with th.no_grad():
            a = self.policy.forward(th.as_tensor(np.asarray(obs)).reshape(-1, self.obs_dim),
                                    None if noise is None else th.as_tensor(noise), deterministic)[0].numpy()
        if not self.discrete:
            a = np.clip(a, self.stack.env.action_low, self.stack.env.action_high)
        return a


def sync_normalization(train_norm, other_norm):
    """ref: vec_env/__init__.py:50-65 — obs_rms and ret_rms are deep-copied, cost_rms is not."""
    other_norm.obs_rms = train_norm.obs_rms.copy()
    other_norm.ret_rms = train_norm.ret_rms.copy()


def sample_from_agent(agent, stack1, rollouts, noise=None):
    """ref: icrl/utils.py:323-357.  1-env stack; records the observation *after* each step next to the
    action that produced it.  noise: optional [total_steps, act] standard normals."""
    assert stack1.num_envs == 1
    orig_obs, obs_l, acts, rews, lens = [], [], [], [], []
    k = 0
    obs = None
    for i in range(rollouts):
        if i == 0:
            obs = stack1.reset()
        done, ep_r, ep_l = False, 0.0, 0
        while not done:
            a = agent.predict(obs, None if noise is None else noise[k:k + 1])
            k += 1
            obs, r, d, _ = stack1.step(a)
            done = bool(d[0])
            obs_l.append(obs[0]); orig_obs.append(stack1.old_obs[0].copy()); acts.append(a[0])
            ep_r += float(r[0]); ep_l += 1
        rews.append(ep_r); lens.append(ep_l)
    return np.array(orig_obs), np.array(obs_l), np.array(acts), np.array(rews), np.array(lens)


def evaluate_policy(agent, stack1, n_eval_episodes=10, noise=None, deterministic=False):
    """ref: common/evaluation.py:10-67."""
    ep_rewards, k, obs = [], 0, None
    for i in range(n_eval_episodes):
        if i == 0:
            obs = stack1.reset()
        done, ep_r = False, 0.0
        while not done:
            a = agent.predict(obs, None if noise is None else noise[k:k + 1], deterministic)
            k += 1
            obs, r, d, _ = stack1.step(a)
            done = bool(d[0]); ep_r += float(r[0])
        ep_rewards.append(ep_r)
    return float(np.mean(ep_rewards)), float(np.std(ep_rewards))


def compute_kl(policy_2, observations, actions, policy_1=None):
    """ref: icrl/utils.py:421-437 on *unnormalised* observations.  QUIRK kept: the reference reads element [1] of
    evaluate_actions(), which for the two-critics policy (policies.py:752-767 -> values, cost_values, log_prob, entropy) is the
    COST VALUE, not the log-probability; the logged `true/forward_kl` / `true/reverse_kl` are therefore
    mean(V_c^{agent_1} - V_c^{agent_2}) (pinned by tests/golden/g8_icrl_lgw.npz)."""
    o = th.tensor(np.asarray(observations), dtype=th.float32)
    a = th.tensor(np.asarray(actions), dtype=th.float32)
    with th.no_grad():
        kl = -policy_2.evaluate_actions(o, a)[1]
        if policy_1 is not None:
            kl = kl + policy_1.evaluate_actions(o, a)[1]
    return (kl.sum() / o.shape[0]).item()


def make_stack(n_envs, kind, seed, *, training=True, norm_reward=True, norm_cost=True, norm_obs=True,
               cost_fn=None, wall_terminate=False, broken=False, reward_gamma=0.99, cost_gamma=0.99):
    if kind in ("lgw", "clgw"):
        env = LapGridVecEnv(n_envs, constrained=(kind == "clgw"))
    else:
        env = SynthVecEnv(n_envs, kind, seed, wall_terminate=wall_terminate, broken=broken)
    norm = stats.NormState(n_envs, env.obs_dim, training=training, norm_obs=norm_obs, norm_reward=norm_reward,
                           norm_cost=norm_cost, reward_gamma=reward_gamma, cost_gamma=cost_gamma)
    return EnvStack(env, norm, cost_fn)


ENV_KINDS = {  # reference gym id -> (oracle env kind, early termination, broken); custom_envs/__init__.py:43-57,194-224,357-370
    "HCWithPos-v0": ("hc", False, False), "HCWithPosTest-v0": ("hc", True, False),
    "AntWall-v0": ("ant", False, False), "AntWallTest-v0": ("ant", True, False),
    "AntWallBroken-v0": ("ant", False, True), "AntWallBrokenTest-v0": ("ant", True, True),
    "LGW-v0": ("lgw", False, False), "CLGW-v0": ("clgw", True, False)}

PORT_DEFAULTS = dict(   # the reference's flag names and parser defaults (icrl/icrl.py:316-417)
    train_env_id="HCWithPos-v0", eval_env_id="HCWithPosTest-v0", num_threads=5, seed=0, n_steps=2048, batch_size=64, n_epochs=10,
    learning_rate=3e-4, reward_gamma=0.99, reward_gae_lambda=0.95, cost_gamma=0.99, cost_gae_lambda=0.95, clip_range=0.2,
    ent_coef=0.0, reward_vf_coef=0.5, cost_vf_coef=0.5, max_grad_norm=0.5, target_kl=None, penalty_initial_value=1.0,
    penalty_learning_rate=0.1, budget=0.0, cn_layers=(64, 64), cn_learning_rate=3e-4, anneal_clr_by_factor=1.0, cn_reg_coeff=0.0,
    no_importance_sampling=False, per_step_importance_sampling=False, cn_target_kl_old_new=10, cn_target_kl_new_old=10,
    cn_batch_size=None, train_gail_lambda=False, cn_normalize=False, backward_iters=10, forward_timesteps=1000000, n_iters=100,
    expert_rollouts=20, clip_obs=20, cn_eps=1e-5, dont_normalize_obs=False, dont_normalize_reward=False,
    dont_normalize_cost=False, warmup_timesteps=None, reset_policy=False, factored=False,
    policy_layers=(64, 64), reward_vf_layers=(64, 64), cost_vf_layers=(64, 64), shared_layers=None)      # -pl / -rvl / -cvl / -sl (icrl/utils.py:636-655)


def true_cost(eval_env_id, orig_obs, acts):
    """ref: icrl/true_constraint_net.py:11-55,104-111."""
    if eval_env_id == "CLGW-v0":
        return float(np.mean(np.asarray(acts).reshape(len(acts), -1)[:, 0] == 1))
    if eval_env_id.endswith("Test-v0"):
        return float(np.mean(orig_obs[..., 0] <= -3))
    return 0.0


def icrl_port(cfg, expert_obs, expert_acs, expert_policy_sd=None, n_iters=None, log=None, streams=None, init=None):
    """The ICRL outer loop (ref: icrl/icrl.py:45-304), CPU port.
    cfg: dict with the reference's flag names (PORT_DEFAULTS).  streams: an oracle.streams object (teacher forcing) or None
    (the global numpy / torch generators, like the reference).  init: optional dict(policy=state_dict, cn=state_dict) of
    initial weights.  Returns (per-iteration metrics list, env_steps, seconds, dict(agent, cn, train))."""
    c = dict(PORT_DEFAULTS)
    c.update(cfg)
    n_iters = c["n_iters"] if n_iters is None else n_iters
    kind, _, broken = ENV_KINDS[c["train_env_id"]]
    ekind, ewall, ebroken = ENV_KINDS[c["eval_env_id"]]
    discrete = kind in ("lgw", "clgw")
    nobs, nrew, ncost = not c["dont_normalize_obs"], not c["dont_normalize_reward"], not c["dont_normalize_cost"]
    train = make_stack(c["num_threads"], kind, c["seed"], norm_obs=nobs, norm_reward=nrew, norm_cost=ncost, broken=broken,
                       reward_gamma=c["reward_gamma"], cost_gamma=c["cost_gamma"])
    sampling = make_stack(1, kind, c["seed"], training=False, norm_obs=nobs, norm_reward=False, norm_cost=False, broken=broken)
    evalst = make_stack(1, ekind, c["seed"], training=False, norm_obs=nobs, norm_reward=False, norm_cost=False,
                        wall_terminate=ewall, broken=ebroken)
    env = train.env
    agent = PortAgent(train, n_steps=c["n_steps"], batch_size=c["batch_size"], n_epochs=c["n_epochs"],
                      learning_rate=c["learning_rate"], reward_gamma=c["reward_gamma"],
                      reward_gae_lambda=c["reward_gae_lambda"], cost_gamma=c["cost_gamma"],
                      cost_gae_lambda=c["cost_gae_lambda"], clip_range=c["clip_range"], ent_coef=c["ent_coef"],
                      reward_vf_coef=c["reward_vf_coef"], cost_vf_coef=c["cost_vf_coef"], max_grad_norm=c["max_grad_norm"],
                      target_kl=c["target_kl"], penalty_initial_value=c["penalty_initial_value"],
                      penalty_learning_rate=c["penalty_learning_rate"], budget=c["budget"], seed=c["seed"], discrete=discrete,
                      hidden=dict(policy_net=tuple(c["policy_layers"]), value_net=tuple(c["reward_vf_layers"]),
                                  cost_value_net=tuple(c["cost_vf_layers"])), shared=tuple(c.get("shared_layers") or ()))
    # NB the reference builds the constraint net BEFORE the agent (icrl.py:88-117 vs :139-178) but the agent's constructor
    # re-seeds every generator (common/utils.py:23-39), so the construction order only matters for the net's own draw.
    cn = CostNet(env.obs_dim, env.act_dim, c["cn_layers"], discrete, None, None, c["clip_obs"],
                 env.action_low, env.action_high, c["cn_eps"])
    if c["cn_normalize"]:
        cn.obs_mean, cn.obs_var = np.zeros(env.obs_dim), np.ones(env.obs_dim)
    if init is not None:
        agent.policy.load_state_dict(init["policy"]); cn.load_state_dict(init["cn"])
    lr_sched = lambda x: (c["anneal_clr_by_factor"] ** (c["n_iters"] * (1 - x))) * c["cn_learning_rate"]
    cn_opt = th.optim.Adam(cn.parameters(), lr=lr_sched(1), eps=1e-5)
    train.cost_fn = cn.cost_function
    expert_policy = None
    if expert_policy_sd is not None:
        expert_policy = TwoCriticPolicy(env.obs_dim, env.act_dim, discrete=discrete)
        expert_policy.load_state_dict(expert_policy_sd)
    A = 1 if discrete else env.act_dim
    out, steps, t0 = [], 0, time.time()
    best = {"reward": -np.inf, "cost": np.inf, "fkl": np.inf, "rkl": np.inf}
    if c["warmup_timesteps"] is not None:
        raise NotImplementedError("warm-up with null_cost is exercised through PortAgent directly")
    for itr in range(n_iters):
        progress = 1 - float(itr) / float(c["n_iters"])
        t_it = time.time()
        agent.learn(c["forward_timesteps"], streams=streams)
        t_fwd = time.time() - t_it
        steps += agent.num_timesteps
        fwd = dict(agent.logs)
        sync_normalization(train.norm, sampling.norm)
        agent.stack, keep = sampling, agent.stack          # predict() clips with the sampling env's action box
        noise = None if streams is None else streams.sample_noise(c["expert_rollouts"] * sampling.env.max_steps, A)
        orig_obs, obs, acts, rews, lens = sample_from_agent(agent, sampling, c["expert_rollouts"], noise)
        for g in cn_opt.param_groups:
            g["lr"] = lr_sched(progress)
        if c["cn_normalize"]:                              # ref: icrl.py:232-236, constraint_net.py:148-153
            cn.obs_mean, cn.obs_var = sampling.norm.obs_rms.mean.copy(), sampling.norm.obs_rms.var.copy()
        nominal = cn.prepare(orig_obs, acts)
        expert_data = cn.prepare(expert_obs, expert_acs)   # re-prepared on every train() (constraint_net.py:156)
        nois = c["no_importance_sampling"] or c["train_gail_lambda"]
        bw = cn_train(cn, cn_opt, c["backward_iters"], nominal, expert_data, lens, reg_coeff=c["cn_reg_coeff"],
                      importance_sampling=not nois, per_step=c["per_step_importance_sampling"],
                      target_kl_old_new=c["cn_target_kl_old_new"], target_kl_new_old=c["cn_target_kl_new_old"],
                      eps=c["cn_eps"], gail=c["train_gail_lambda"], batch_size=c["cn_batch_size"], factored=c["factored"])
        tc = true_cost(c["eval_env_id"], orig_obs, acts)
        sync_normalization(train.norm, evalst.norm)
        agent.stack = evalst
        enoise = None if streams is None else streams.eval_noise(10 * evalst.env.max_steps, A)
        rew_mean, rew_std = evaluate_policy(agent, evalst, 10, enoise)
        agent.stack = keep
        best["reward"], best["cost"] = max(best["reward"], rew_mean), min(best["cost"], tc)
        m = {"iteration": itr, "timesteps": steps, "true/reward": rew_mean, "true/reward_std": rew_std, "true/cost": tc,
             "true/samples_behind": float(np.mean(orig_obs[..., 0] < -3)), "true/samples_infront": float(np.mean(orig_obs[..., 0] > 3)),
             "best_true/best_reward": best["reward"], "best_true/best_cost": best["cost"]}
        if expert_policy is not None:
            m["true/forward_kl"] = compute_kl(agent.policy, expert_obs, expert_acs, expert_policy)
            m["true/reverse_kl"] = compute_kl(expert_policy, orig_obs, acts, agent.policy)
            best["fkl"], best["rkl"] = min(best["fkl"], m["true/forward_kl"]), min(best["rkl"], m["true/reverse_kl"])      # ref: icrl.py:276-279
            m["best_true/best_forward_kl"], m["best_true/best_reverse_kl"] = best["fkl"], best["rkl"]
        m.update({k.replace("train/", "forward/"): v for k, v in fwd.items() if k.startswith(("train/", "time/"))})      # ref: icrl.py:299
        m.update(bw)
        m["time/forward_s"], m["time/rest_s"] = t_fwd, time.time() - t_it - t_fwd      # wall clock (bench.py's cpu_baseline)
        out.append(m)
        if log:
            log(m)
    return out, steps, time.time() - t0, dict(agent=agent, cn=cn, train=train, sampling=sampling)
