"""PPOLagrangian — PPO-clip with a Lagrangian cost term and two critics, device-resident.

ref: stable_baselines3/ppo_lag/ppo_lag.py:17-365 (PPOLagrangian: __init__, _setup_model, train, learn)
     stable_baselines3/common/on_policy_algorithm.py:258-497 (OnPolicyWithCostAlgorithm: collect_rollouts, learn)
     stable_baselines3/common/base_class.py:303-366 (_setup_learn: env reset on every learn())

The host keeps the reference's control flow and attribute names (num_timesteps, rollout_buffer, dual, policy, ...);
the work is enqueued as kernels of libicrl_hip.so:
  collect_rollouts -> icrl_rollout_collect (ONE persistent launch for all T steps, or 2 launches per env step, + dual GAE;
                      no host sync)
  train            -> icrl_ppo_lag_train   (ONE persistent launch for all epochs x minibatches); the dual variable is one
                      float32 scalar updated on the host (dual_variable.py)
"""
import time

import numpy as np
import torch

from . import _lib, logger, spaces
from .buffers import RolloutBufferWithCost
from .dual_variable import DualVariable, PIDLagrangian
from .policies import ActorTwoCriticsPolicy
from .structs import AgentT, PpoHyperT, p
from .vec_env import HipSynthVecEnv, VecCostWrapper, VecEnv, VecNormalize, VecNormalizeWithCost


def _const_fn(v):
    return v if callable(v) else (lambda _progress: v)


class _SpacesOnlyEnv(VecEnv):
    """what a loaded agent holds when no env was given (reference: env=None, prediction only)."""

    def reset(self):
        raise RuntimeError("this agent was loaded without an environment")

    step_async = step_wait = reset


class PPOLagrangian:
    def __init__(self, policy, env, algo_type="lagrangian", learning_rate=3e-4, n_steps=2048, batch_size=64, n_epochs=10,
                 reward_gamma=0.99, reward_gae_lambda=0.95, cost_gamma=0.99, cost_gae_lambda=0.95, clip_range=0.2,
                 clip_range_reward_vf=None, clip_range_cost_vf=None, ent_coef=0.0, reward_vf_coef=0.5, cost_vf_coef=0.5,
                 max_grad_norm=0.5, use_sde=False, sde_sample_freq=-1, target_kl=None, penalty_initial_value=1,
                 penalty_learning_rate=0.01, penalty_min_value=None, update_penalty_after=1, budget=0.,
                 tensorboard_log=None, create_eval_env=False, pid_kwargs=None, policy_kwargs=None, verbose=0, seed=None,
                 device="cuda", _init_setup_model=True, action_noise="device", permutation="numpy", streams=None):
        if use_sde:
            raise NotImplementedError("gSDE is outside the hot path (no BASELINE config uses it)")
        if policy not in ("TwoCriticsMlpPolicy", ActorTwoCriticsPolicy):
            raise NotImplementedError(f"policy {policy!r}: only TwoCriticsMlpPolicy is on the ICRL hot path")
        self.env: VecEnv = env
        self.n_envs = env.num_envs
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.device = torch.device("cuda" if device in ("auto", "cpu", None) else device)
        self.algo_type, self.learning_rate, self.n_steps = algo_type, learning_rate, n_steps
        self.batch_size, self.n_epochs = batch_size, n_epochs
        self.reward_gamma, self.reward_gae_lambda = reward_gamma, reward_gae_lambda
        self.cost_gamma, self.cost_gae_lambda = cost_gamma, cost_gae_lambda
        self.clip_range, self.clip_range_reward_vf, self.clip_range_cost_vf = clip_range, clip_range_reward_vf, clip_range_cost_vf
        self.ent_coef, self.reward_vf_coef, self.cost_vf_coef, self.max_grad_norm = ent_coef, reward_vf_coef, cost_vf_coef, max_grad_norm
        self.target_kl = target_kl
        self.penalty_initial_value, self.penalty_learning_rate = penalty_initial_value, penalty_learning_rate
        self.penalty_min_value, self.update_penalty_after, self.budget = penalty_min_value, update_penalty_after, budget
        self.pid_kwargs, self.policy_kwargs = pid_kwargs, {} if policy_kwargs is None else policy_kwargs
        self.verbose, self.seed = verbose, seed
        # how the two random streams of the hot path are produced (both are explicit kernel inputs):
        #   action_noise: "device" (torch.randn on the GPU) | "torch_cpu" (global CPU generator, one call per step: the
        #                 reference's stream) | callable(T, N, A) -> array
        #   permutation:  "numpy" (np.random.permutation per epoch: the reference's stream) | "device" | callable
        #   streams:      optional object with rollout_noise(T, N, A), permutation(epoch, n), consumed(executed_epochs): every
        #                 draw of learn() comes from it (teacher forcing against the reference / the CPU oracle in the tests)
        self.action_noise, self.permutation, self.streams = action_noise, permutation, streams
        self.num_timesteps, self._n_updates, self._total_timesteps = 0, 0, 0
        self._current_progress_remaining = 1
        self._last_obs = self._last_original_obs = self._last_dones = None
        self.start_time = None
        self._vec_normalize_env = env if isinstance(env, VecNormalize) else None
        if _init_setup_model:
            self._setup_model()

    # ref: on_policy_algorithm.py:313-338, ppo_lag.py:147-175, common/utils.py:23-39
    def _setup_model(self):
        self.lr_schedule = _const_fn(self.learning_rate)
        if self.seed is not None:
            import random
            random.seed(self.seed); np.random.seed(self.seed); torch.manual_seed(self.seed)
            self.action_space.seed(self.seed)
            self.env.seed(self.seed)
        self.rollout_buffer = RolloutBufferWithCost(self.n_steps, self.observation_space, self.action_space, self.device,
                                                    self.reward_gamma, self.reward_gae_lambda, self.cost_gamma,
                                                    self.cost_gae_lambda, n_envs=self.n_envs)
        self.policy = ActorTwoCriticsPolicy(self.observation_space, self.action_space, self.lr_schedule,
                                            device=self.device, **self.policy_kwargs)
        if self.algo_type == "lagrangian":
            self.dual = DualVariable(self.budget, self.penalty_learning_rate, self.penalty_initial_value, self.penalty_min_value)
        elif self.algo_type == "pidlagrangian":
            self.dual = PIDLagrangian(**{k: self.pid_kwargs[k] for k in ("alpha", "penalty_init", "Kp", "Ki", "Kd", "pid_delay",
                                                                         "delta_p_ema_alpha", "delta_d_ema_alpha")})
        else:
            raise ValueError("Unrecognized value for argument 'algo_type' in PPOLagrangian")
        self.clip_range = _const_fn(self.clip_range)
        if self.clip_range_reward_vf is not None:
            self.clip_range_reward_vf = _const_fn(self.clip_range_reward_vf)
        if self.clip_range_cost_vf is not None:
            self.clip_range_cost_vf = _const_fn(self.clip_range_cost_vf)
        N, dev = self.n_envs, self.device
        A = 1 if isinstance(self.action_space, spaces.Discrete) else self.action_space.shape[0]
        self._ag = dict(last_dones=torch.zeros(N, dtype=torch.uint8, device=dev), raw_rew=torch.zeros(N, dtype=torch.float64, device=dev),
                        raw_cost=torch.zeros(N, device=dev), dones=torch.zeros(N, dtype=torch.uint8, device=dev),
                        last_v_r=torch.zeros(N, device=dev), last_v_c=torch.zeros(N, device=dev),
                        act_clipped=torch.zeros(N, A, device=dev), status=torch.zeros(1, dtype=torch.int32, device=dev),
                        # exchange workspace of the persistent rollout: ICRL_ROLLOUT_WS_BYTES(N, obs)
                        xch_ws=torch.zeros((48 * N * self.observation_space.shape[0] + 96 * N + 64 * self.observation_space.shape[0] + 2048) // 8 + 1,
                                           dtype=torch.int64, device=dev))
        if isinstance(self.action_space, spaces.Box):
            self._alow = torch.as_tensor(self.action_space.low, device=dev).float().contiguous()
            self._ahigh = torch.as_tensor(self.action_space.high, device=dev).float().contiguous()
        else:
            self._alow = self._ahigh = None

    # ---- env-chain introspection: is this the device-native stack the fused rollout handles? ---------------------------
    def _fused_chain(self):
        env = self.env
        if not isinstance(env, VecNormalizeWithCost):
            return None
        # (a policy / constraint net of the generic-shape path — layers above 64 units, shared trunk, other depths — takes the same entry
        # point; the library then runs the rollout as one persistent launch around the table-driven forward — or, for a constraint net beyond
        # two layers of 64 units, the reference's per-step loop as four launches per step —, csrc/rollout.hip)
        cw = env.venv
        if isinstance(cw, HipSynthVecEnv):           # no cost wrapper in the chain (the GAIL baseline, icrl/gail.py:50-59): costs are 0
            return env, None, cw
        if not isinstance(cw, VecCostWrapper) or not isinstance(cw.venv, HipSynthVecEnv) or cw.constraint_net() is None:
            return None
        return env, cw, cw.venv

    # ---- noise / permutation streams -------------------------------------------------------------------------------------
    def _draw_action_noise(self, T):
        N = self.n_envs
        disc = isinstance(self.action_space, spaces.Discrete)
        shape = (T, N) if disc else (T, N, self.action_space.shape[0])
        if self.streams is not None:
            noise = self.streams.rollout_noise(T, N, 1 if disc else shape[2])
            if not torch.is_tensor(noise):
                noise = torch.as_tensor(np.asarray(noise, np.float32), device=self.device)
            return noise.to(device=self.device, dtype=torch.float32).reshape(shape).contiguous()
        if callable(self.action_noise):
            return torch.as_tensor(np.asarray(self.action_noise(*shape), np.float32), device=self.device).reshape(shape).contiguous()
        if self.action_noise == "torch_cpu":        # one generator call per env step, like Normal.rsample in the reference
            rows = [torch.rand(shape[1:]) if disc else torch.randn(shape[1:]) for _ in range(T)]
            return torch.stack(rows).to(self.device).contiguous()
        return (torch.rand(shape, device=self.device) if disc else torch.randn(shape, device=self.device)).contiguous()

    # ---- rollout ---------------------------------------------------------------------------------------------------------
    def collect_rollouts(self, env, callback, rollout_buffer, n_rollout_steps, cost_function="cost", noise=None):
        """ref: on_policy_algorithm.py:340-421."""
        assert self._last_obs is not None, "No previous observation was provided"
        if not self._fused_rollout_ok(cost_function, n_rollout_steps, rollout_buffer):
            return self._collect_rollouts_stepped(env, callback, rollout_buffer, n_rollout_steps, cost_function, noise)
        job = self._rollout_begin(callback, rollout_buffer, n_rollout_steps, noise)
        self._rollout_launch(job)
        return self._rollout_end(job, env, callback, rollout_buffer, n_rollout_steps)

    def _fused_rollout_ok(self, cost_function, n_rollout_steps, rollout_buffer):
        return self._fused_chain() is not None and isinstance(cost_function, str) and n_rollout_steps == rollout_buffer.buffer_size

    # The fused rollout in three pieces, so that several runs sharing a GPU (icrl_amd/seed_batch.py) can put their launches into
    # ONE grid: _rollout_begin (host state + descriptors), the launch (single: _rollout_launch; batched: seed_batch), _rollout_end.
    def _rollout_begin(self, callback, rollout_buffer, n_rollout_steps, noise=None, zero_buffer=True):
        if zero_buffer:
            rollout_buffer.reset()
        else:            # the launch overwrites every row of all 16 arrays: the zeroing is invisible
            rollout_buffer.pos, rollout_buffer.full, rollout_buffer.generator_ready = 0, False, False
        if callback is not None:
            callback.on_rollout_start()
        nenv, cw, senv = self._fused_chain()
        if noise is None:
            noise = self._draw_action_noise(n_rollout_steps)
        e, nm, pol, buf = senv.struct(), nenv.struct(), self.policy.struct(), rollout_buffer.struct()
        cn = cw.constraint_net().struct() if cw is not None else None
        ag = AgentT(p(self._last_obs), p(self._ag["last_dones"]), p(self._ag["raw_rew"]), p(self._ag["raw_cost"]), p(self._ag["dones"]),
                    p(self._ag["last_v_r"]), p(self._ag["last_v_c"]), p(self._ag["act_clipped"]), p(self._ag["status"]),
                    p(self._ag["xch_ws"]), self._ag["xch_ws"].numel() * 8)
        timed = getattr(self, "gae_events", None) is not None
        flags = int(not timed) | {"steps": 2, "wide": 16, "multi": 32}.get(getattr(self, "rollout_kernel", "auto"), 0) | ((4 if getattr(self, "_wide_prof_flag", 1) == 1 else 8) if getattr(self, "profile_phases", 0) else 0)
        return dict(env=e, nm=nm, pol=pol, cn=cn, buf=buf, ag=ag, noise=noise, flags=flags, timed=timed, chain=(nenv, cw, senv))

    def _rollout_launch(self, job):
        b = _lib.byref
        rev = None
        if getattr(self, "rollout_events", None) is not None:      # bench.py: events around the rollout launch on its stream
            rev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            rev[0].record()
        cn = job["cn"]
        _lib.check(_lib.lib().icrl_rollout_collect_ex(b(job["env"]), b(job["nm"]), b(job["pol"]), b(cn) if cn is not None else None, b(job["buf"]), b(job["ag"]),
                                                      p(job["noise"]), p(self._alow), p(self._ahigh),
                                                      float(self.reward_gamma), float(self.reward_gae_lambda), float(self.cost_gamma),
                                                      float(self.cost_gae_lambda), job["flags"], _lib.current_stream()),
                   "icrl_rollout_collect")
        if rev is not None:
            rev[1].record()
            self.rollout_events.append(rev)

    def _rollout_end(self, job, env, callback, rollout_buffer, n_rollout_steps):
        nenv, cw, senv = job["chain"]
        if job["timed"]:   # bench.py: the same GAE launch, bracketed by events on the stream it runs on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rollout_buffer.compute_returns_and_advantage(self._ag["last_v_r"], self._ag["last_v_c"], self._ag["last_dones"])
            e1.record()
            self.gae_events.append((e0, e1))
        self._keepalive = (job["noise"],)
        rollout_buffer.pos, rollout_buffer.full = n_rollout_steps, True
        self.num_timesteps += env.num_envs * n_rollout_steps
        # wrapper-visible "last step" state, as the reference leaves it
        nenv.old_obs, nenv.old_reward, nenv.old_cost = senv.s, self._ag["raw_rew"], self._ag["raw_cost"]
        if cw is not None:
            cw.previous_obs = senv.s
        self._last_original_obs = senv.s
        self._last_dones = self._ag["last_dones"]
        if callback is not None:
            if hasattr(callback, "on_steps") and callback.on_steps(n_rollout_steps) is False:     # n_steps on_step() calls at once
                return False
            callback.on_rollout_end()
        return True

    def check_rollout_status(self):
        """raise if the persistent rollout kernel reported a timed-out exchange (icrl_agent_t.status): its buffer rows and
        running moments are then invalid.  One 4-byte read; train() calls it next to its own statistics read-back."""
        both = torch.cat([self._ag["status"].flatten()[:1].to(torch.int32), self.rollout_buffer.gae_status]).cpu()
        if int(both[1]) != 0:
            self.rollout_buffer.check_gae_status()
        if int(both[0]) != 0:
            self._ag["status"].zero_()
            raise RuntimeError("icrl_rollout_collect: inter-workgroup exchange timed out (a workgroup of the persistent "
                               "rollout was not resident); rollout buffer and normaliser statistics are invalid")

    def _collect_rollouts_stepped(self, env, callback, rollout_buffer, n_rollout_steps, cost_function, noise=None):
        """The reference's per-step loop, kept for everything the fused launch does not cover: a callable `cost_function`
        (warm-up with null_cost, icrl/icrl.py:187-193; costs are then evaluated on the observation AFTER the step and are not
        normalised, on_policy_algorithm.py:392-394), an env cost that is an arbitrary Python function (cpg with the analytic
        cost), partial rollouts.  One launch per call of the fine-grained C-ABI entry points; not the benchmarked path."""
        rollout_buffer.reset()
        if callback is not None:
            callback.on_rollout_start()
        is_box = isinstance(self.action_space, spaces.Box)
        norm_env = env if isinstance(env, VecNormalizeWithCost) else None
        v_r = v_c = dones = None
        if noise is None and self.streams is not None:      # teacher-forced streams serve this path like the fused one
            noise = self._draw_action_noise(n_rollout_steps)
        for t in range(n_rollout_steps):
            actions, v_r, v_c, log_probs = self.policy.forward(self._last_obs, noise=None if noise is None else noise[t])
            clipped = self.policy.last_clipped if is_box else actions
            new_obs, rewards, dones, infos = env.step(clipped)
            orig_obs = env.get_original_obs() if norm_env is not None else new_obs
            if isinstance(cost_function, str):
                costs = infos.batch.get(cost_function) if hasattr(infos, "batch") else None
                if costs is None:
                    costs = torch.zeros(env.num_envs, device=self.device)
                orig_costs = env.get_original_cost() if norm_env is not None and norm_env.old_cost is not None else costs
            else:    # numpy in / numpy out like the reference's callables (true_constraint_net.py)
                costs = np.asarray(cost_function(orig_obs.cpu().numpy().copy(), clipped.cpu().numpy()), dtype=np.float32)
                orig_costs = costs
            self.num_timesteps += env.num_envs
            if callback is not None and hasattr(callback, "on_step"):
                if hasattr(callback, "update_locals"):
                    callback.update_locals(locals())
                if callback.on_step() is False:
                    return False
            rollout_buffer.add(self._last_obs, self._last_original_obs, new_obs, orig_obs, actions, rewards, costs, orig_costs,
                               self._last_dones, v_r, v_c, log_probs)
            self._last_obs, self._last_original_obs, self._last_dones = new_obs.contiguous(), orig_obs, dones
        self._ag["last_dones"].copy_(torch.as_tensor(dones, device=self.device).to(torch.uint8))
        self._last_dones = self._ag["last_dones"]
        rollout_buffer.compute_returns_and_advantage(v_r, v_c, dones)
        if callback is not None:
            callback.on_rollout_end()
        return True

    def _setup_learn(self, total_timesteps, reset_num_timesteps=True):
        """ref: base_class.py:303-366."""
        self.start_time = time.time()
        if reset_num_timesteps:
            self.num_timesteps = 0
        else:
            total_timesteps += self.num_timesteps
        self._total_timesteps = total_timesteps
        if reset_num_timesteps or self._last_obs is None:
            self._last_obs = self.env.reset().contiguous()
            self._ag["last_dones"].zero_()
            self._last_dones = self._ag["last_dones"]
            self._last_original_obs = self._vec_normalize_env.get_original_obs() if self._vec_normalize_env is not None else self._last_obs
        return total_timesteps

    def _training_infos(self, itr):
        """ref: on_policy_algorithm.py:452-457 (Monitor's `rollout/ep_*` episode statistics are not kept: the envs are device-resident)."""
        elapsed = max(time.time() - self.start_time, 1e-9)
        logger.record("time/iterations", itr)
        logger.record("time/fps", int(self.num_timesteps / elapsed))
        logger.record("time/time_elapsed", int(elapsed))
        logger.record("time/total_timesteps", self.num_timesteps)

    def learn(self, total_timesteps, cost_function="cost", callback=None, log_interval=1, eval_env=None, eval_freq=-1,
              n_eval_episodes=5, tb_log_name="PPOLagrangian", eval_log_path=None, reset_num_timesteps=True):
        """ref: on_policy_algorithm.py:430-492."""
        iteration = 0
        total_timesteps = self._setup_learn(total_timesteps, reset_num_timesteps)
        if callback is not None:
            callback.init_callback(self)
            callback.on_training_start(locals(), globals())
        while self.num_timesteps < total_timesteps:
            if self.collect_rollouts(self.env, callback, self.rollout_buffer, self.n_steps, cost_function) is False:
                break
            iteration += 1
            self._current_progress_remaining = 1.0 - float(self.num_timesteps) / float(total_timesteps)
            if log_interval is not None and iteration % log_interval == 0:
                self._training_infos(iteration)
                logger.dump(step=self.num_timesteps)
            self.train()
        self._training_infos(iteration + 1)       # (the reference's closing call: what icrl() scrapes holds iterations + 1, on_policy_algorithm.py:488)
        if callback is not None:
            callback.on_training_end()
        return self

    # ---- persistence (ref: common/base_class.py save / load, common/save_util.py:284-418) ---------------------------------
    def save(self, path):
        """SB3-style archive the REFERENCE can load (ref: base_class.py:647-692 save, save_util.py:72-119 data_to_json, :284-322
        save_to_zip_file): `<path>.zip` with `policy.pth` (state dict under the reference's parameter names), `policy.optimizer.pth`
        (torch Adam layout), `pytorch_variables.pth` (empty, as the reference writes it) and `data` = the JSON of the constructor
        attributes, in which `policy_class`, `observation_space` and `action_space` are pickles BY REFERENCE to
        stable_baselines3.common.policies.ActorTwoCriticsPolicy / gym.spaces.box.Box / gym.spaces.discrete.Discrete (utils.sb3_*_entry:
        no gym needed to write them).  The reference's `PPOLagrangian.load(path)` (base_class.py:564-645) rebuilds the agent from it —
        checked in the build container by oracle/verify_agent_archive.py.  Schedules are stored by their current value (the reference
        re-wraps floats in `_setup_model`); what only this build reads (the dual variable's state, the Adam step count) sits in a
        `dual_state.json` member the reference ignores."""
        import io, json, zipfile
        from . import utils as U
        path = str(path)
        path = path if path.endswith(".zip") else path + ".zip"
        def blob(obj):
            b = io.BytesIO(); torch.save(obj, b); return b.getvalue()
        data = {k: getattr(self, k) for k in ("n_steps", "batch_size", "n_epochs", "reward_gamma", "reward_gae_lambda", "cost_gamma",
                                               "cost_gae_lambda", "ent_coef", "reward_vf_coef", "cost_vf_coef", "max_grad_norm",
                                               "target_kl", "n_envs", "num_timesteps", "_n_updates", "seed")}
        num = lambda v: None if v is None else float(v(1.0) if callable(v) else v)
        data.update(learning_rate=num(self.learning_rate), clip_range=num(self.clip_range), clip_range_reward_vf=num(self.clip_range_reward_vf),
                    clip_range_cost_vf=num(self.clip_range_cost_vf), algo_type=self.algo_type, budget=float(self.budget),
                    penalty_initial_value=float(self.penalty_initial_value), penalty_learning_rate=float(self.penalty_learning_rate),
                    penalty_min_value=self.penalty_min_value, update_penalty_after=self.update_penalty_after, pid_kwargs=self.pid_kwargs,
                    policy_kwargs=dict(net_arch=[*self.policy.shared, dict(pi=list(self.policy.layers["policy_net"]), vf=list(self.policy.layers["value_net"]),
                                                                           cvf=list(self.policy.layers["cost_value_net"]))]))
        # the rest of what the reference's __init__ leaves in self.__dict__ and its load() / _setup_model() / predict() read
        data.update(verbose=int(getattr(self, "verbose", 0)), use_sde=False, sde_sample_freq=-1, tensorboard_log=None, action_noise=None,
                    _total_timesteps=int(getattr(self, "_total_timesteps", 0) or 0), _episode_num=0, start_time=None,
                    _current_progress_remaining=float(getattr(self, "_current_progress_remaining", 1.0)))
        data.update(policy_class=U.sb3_policy_class_entry(), observation_space=U.sb3_space_entry(self.observation_space),
                    action_space=U.sb3_space_entry(self.action_space))
        def space_exact(sp):      # shape and bounds EXACTLY (the printable `low` / `high` fields of `data` are str(ndarray): numpy elides arrays
            if hasattr(sp, "n"):  # above 1000 elements and prints 8 decimals) — load() prefers this entry; the reference ignores the member
                return dict(kind="discrete", n=int(sp.n))
            return dict(kind="box", shape=[int(x) for x in sp.shape], dtype=str(np.dtype(sp.dtype)),
                        low=np.asarray(sp.low, np.float64).reshape(-1).tolist(), high=np.asarray(sp.high, np.float64).reshape(-1).tolist())
        extra = dict(dual=self.dual.state_dict() if hasattr(self.dual, "state_dict") else {}, adam_step=int(self.policy.adam_step),
                     writer="icrl_amd", spaces=dict(observation=space_exact(self.observation_space), action=space_exact(self.action_space)))
        with zipfile.ZipFile(path, "w") as z:
            z.writestr("data", json.dumps(data, indent=4, default=lambda o: str(o)))
            z.writestr("pytorch_variables.pth", blob({}))
            z.writestr("policy.pth", blob(self.policy.state_dict()))
            z.writestr("policy.optimizer.pth", blob(self.policy.optimizer_state_dict(lr=float(self.lr_schedule(1.0)))))
            z.writestr("dual_state.json", json.dumps(extra))
            z.writestr("_stable_baselines3_version", "0.9.0")
        return path

    @classmethod
    def load(cls, path, env=None, device="auto", **kwargs):
        """ref: common/base_class.py:564-645 + save_util.py:284-418 — rebuild an agent from an archive written by the reference
        (`best_model.zip`) or by save().  As in the reference, `_setup_model()` runs AFTER the stored attributes are restored, so
        the dual variable is re-created: a loaded agent's nu is back at penalty_initial_value.  `env` may be
        None (prediction / evaluate_actions only).  Entries the reference stored as cloud-pickled objects are not un-pickled:
        spaces are rebuilt from their printable fields, the schedules from `learning_rate` / `kwargs` (clip_range defaults to
        0.2 unless passed)."""
        import io, os, zipfile
        from .utils import parse_sb3_data
        path = str(path)
        if not os.path.exists(path) and os.path.exists(path + ".zip"):      # save_util.open_path appends the suffix (save_util.py:215-229)
            path += ".zip"
        with zipfile.ZipFile(path) as z:
            exact = {}
            if "dual_state.json" in z.namelist():      # archives of this build carry their spaces exactly (any size, full precision)
                import json
                exact = json.loads(z.read("dual_state.json")).get("spaces", {})
            raw = json.loads(z.read("data")) if exact else z.read("data")
            if exact:                                  # (then the printable fields are not parsed at all)
                raw = {k: v for k, v in raw.items() if k not in ("observation_space", "action_space")}
            data = parse_sb3_data(raw)
            for key, name in (("observation", "observation_space"), ("action", "action_space")):
                e = exact.get(key)
                if e is not None:
                    data[name] = (spaces.Discrete(int(e["n"])) if e["kind"] == "discrete" else
                                  spaces.Box(np.asarray(e["low"], np.float64).reshape(e["shape"]), np.asarray(e["high"], np.float64).reshape(e["shape"]),
                                             tuple(e["shape"]), np.dtype(e["dtype"]).type))
            # a state dict of tensors: weights_only refuses anything else (an archive is untrusted input; the `data` entry next to it
            # goes through the allow-list unpickler of utils.parse_sb3_data for the same reason)
            sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        # network widths (-sl / -pl / -rvl / -cvl): read off the stored tensors, whoever wrote the archive
        def w(b):      # the Linear layers of a Sequential sit at its even indices (torch_layers.py:183-226)
            idx = sorted(int(k.split(".")[2]) for k in sd if k.startswith(f"mlp_extractor.{b}.") and k.endswith(".weight"))
            if idx != list(range(0, 2 * len(idx), 2)):
                raise NotImplementedError(f"archive: mlp_extractor.{b} has Linear layers at {idx}, not at 0, 2, 4, ... (tanh between them)")
            return [int(sd[f"mlp_extractor.{b}.{i}.weight"].shape[0]) for i in idx]
        net_arch = [*w("shared_net"), dict(pi=w("policy_net"), vf=w("value_net"), cvf=w("cost_value_net"))]
        if "observation_dim" in data and "observation_space" not in data:      # archive written by save() of this build
            o, a = int(data["observation_dim"]), int(data["action_dim"])
            data["observation_space"] = spaces.Box(-np.inf, np.inf, (o,), np.float64)
            data["action_space"] = spaces.Discrete(a) if data.get("discrete") else spaces.Box(-1.0, 1.0, (a,), np.float32)
        if "observation_space" not in data or "action_space" not in data:
            raise KeyError("The observation_space and action_space were not given, can't verify new environments")
        if env is not None:
            if tuple(env.observation_space.shape) != tuple(data["observation_space"].shape):
                raise ValueError(f"Observation spaces do not match: {env.observation_space.shape} != {data['observation_space'].shape}")
        else:
            env = _SpacesOnlyEnv(int(data.get("n_envs", 1)), data["observation_space"], data["action_space"])
        model = cls("TwoCriticsMlpPolicy", env, device=device, _init_setup_model=False)
        skip = {"policy_class", "observation_space", "action_space", "device", "policy_kwargs", "observation_dim", "action_dim",
                "discrete", "policy_class_name", "adam_step", "n_envs"}
        for k, v in data.items():
            if k not in skip:
                setattr(model, k, v)
        for k in ("clip_range", "lr_schedule"):          # stored as pickled closures by the reference
            if callable(getattr(model, k, None)) is False and not isinstance(getattr(model, k, None), (int, float)):
                setattr(model, k, 0.2 if k == "clip_range" else None)
        model.policy_kwargs = dict(net_arch=net_arch)
        model.__dict__.update(kwargs)
        model._setup_model()
        model.load_parameters(path, dual=False)
        return model

    def load_parameters(self, path, dual=True):
        """restore policy weights (+ optimizer moments and dual variable when present) from an archive written by save() or by
        the reference (`best_model.zip`)."""
        import io, zipfile
        with zipfile.ZipFile(str(path)) as z:
            names = set(z.namelist())
            rd = lambda n: torch.load(io.BytesIO(z.read(n)), map_location="cpu", weights_only=False)
            self.policy.load_state_dict(rd("policy.pth"))
            if "policy.optimizer.pth" in names:
                self.policy.load_optimizer_state_dict(rd("policy.optimizer.pth"))
            if dual and hasattr(self.dual, "load_state_dict"):
                import json
                pv = {}
                if "dual_state.json" in names:                    # archives of this build (round 5 on)
                    pv = json.loads(z.read("dual_state.json")).get("dual", {})
                elif "pytorch_variables.pth" in names:            # rounds 2-4 kept it there; the reference's own archives hold {}
                    pv = rd("pytorch_variables.pth")
                if isinstance(pv, dict) and pv:
                    self.dual.load_state_dict(pv)
        return self

    def predict(self, observation, state=None, mask=None, deterministic=False, noise=None):
        return self.policy.predict(observation, state, mask, deterministic, noise)

    # ---- update -----------------------------------------------------------------------------------------------------------
    def _draw_permutations(self, n):
        """[n_epochs, n] int32 on the device.  "numpy": np.random.permutation per epoch — the reference's stream
        (ref: buffers.py:596); the generator is rewound afterwards to what the reference would have consumed (it stops
        drawing once an epoch early-stops), see train()."""
        if self.streams is not None and hasattr(self.streams, "permutations"):      # one batched device-side draw for all epochs
            return self.streams.permutations(self.n_epochs, n).to(device=self.device, dtype=torch.int32).contiguous(), None
        if self.streams is not None or callable(self.permutation):
            draw = self.streams.permutation if self.streams is not None else self.permutation
            perms = [draw(e, n) for e in range(self.n_epochs)]
            if torch.is_tensor(perms[0]):             # device-side streams (icrl_amd/streams.py): no host work, no upload
                return torch.stack(perms).to(device=self.device, dtype=torch.int32).contiguous(), None
            perms = np.stack([np.asarray(q) for q in perms])
            return torch.as_tensor(perms.astype(np.int32), device=self.device).contiguous(), None
        if self.permutation == "device":
            return torch.stack([torch.randperm(n, device=self.device) for _ in range(self.n_epochs)]).to(torch.int32).contiguous(), None
        perms, states = np.empty((self.n_epochs, n), np.int32), []
        for e in range(self.n_epochs):
            perms[e] = np.random.permutation(n)
            states.append(np.random.get_state())             # generator state after e + 1 epochs' draws
        return torch.as_tensor(perms, device=self.device).contiguous(), states

    # Rollouts of this many rows or more: the reference's per-epoch np.random.permutation (buffers.py:596; 2-8 ms per epoch of host
    # time at 0.5-1 M rows, 20 epochs in the AntWall configurations) is drawn EPOCH BY EPOCH beside the running update instead of all
    # epochs up front — see _train_epochwise.  Below it the draws hide under the asynchronous rollout launch anyway.
    LAZY_PERM_ROWS = 262144

    def train(self, perms=None):
        """ref: ppo_lag.py:177-338."""
        rb = self.rollout_buffer
        if (perms is None and self.streams is None and not callable(self.permutation) and self.permutation != "device" and self.n_epochs > 1
                and rb.buffer_size * rb.n_envs >= self.LAZY_PERM_ROWS):
            return self._train_epochwise()
        job = self._train_begin(perms)
        self._train_launch(job)
        self._train_end(job)

    def _train_epochwise(self):
        """train() as one launch PER EPOCH, the next epoch's permutation drawn on the host while the current epoch runs on the device.
        The reference draws np.random.permutation inside its epoch loop and stops drawing when the target-KL test ends the loop
        (ppo_lag.py:203-299); drawing all n_epochs permutations before a single launch costs n_epochs x (2-8 ms) of host time at
        0.5-1 M rows — 55 ms of a 162 ms outer iteration of BASELINE configs[2], 165 of 358 ms of configs[4], nearly all of it for
        epochs the early stop never runs.  Here: draw(0), launch(0); then for every further epoch draw(e) beside the running epoch
        e - 1, wait for that epoch's statistics, stop if it stopped, launch(e).  At most ONE permutation is drawn in vain, and the
        generator is put back to where the reference leaves it.  Same permutations, same minibatches, same arithmetic: parameters,
        moments and step counter are bit-identical to the single launch (they round-trip through device memory in fp32 between
        launches); the logged SUMS are added up per epoch on the host instead of in one fp32 chain (~1e-7 relative)."""
        rb, pol, dev = self.rollout_buffer, self.policy, self.device
        n, E = rb.buffer_size * rb.n_envs, self.n_epochs
        if getattr(self, "_lazy_perm", None) is None or tuple(self._lazy_perm.shape) != (E, n):
            self._lazy_perm = torch.empty((E, n), dtype=torch.int32, device=dev)
            self._lazy_pin = torch.empty((2, n), dtype=torch.int32).pin_memory()
            self._lazy_stats = torch.empty((E, 33), dtype=torch.float32).pin_memory()
        job = self._train_begin(device_perms=self._lazy_perm)
        ws, b = self._train_ws, _lib.byref      # (no placement calibration in this form — it times whole updates; a later single-launch train() still makes it)
        hp = PpoHyperT.from_buffer_copy(job["hp"]); hp.n_epochs = 1
        states, events = [], []

        def draw(e):
            self._lazy_pin[e % 2].numpy()[:] = np.random.permutation(n)
            states.append(np.random.get_state())                       # generator state after e + 1 draws
            self._lazy_perm[e].copy_(self._lazy_pin[e % 2], non_blocking=True)

        def launch(e):
            _lib.check(_lib.lib().icrl_ppo_lag_train(b(job["ps"]), p(pol.exp_avg), p(pol.exp_avg_sq), p(ws["t"]), b(job["bs"]), p(self._lazy_perm[e]), p(ws["nu"]),
                                                     b(hp), p(ws["stats"]), p(ws["sync"]), _lib.current_stream()), "icrl_ppo_lag_train")
            self._lazy_stats[e].copy_(ws["stats"][:33], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(); events.append(ev)

        ev = None
        if getattr(self, "train_events", None) is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        draw(0); launch(0)
        executed = 1
        for e in range(1, E):
            draw(e)                                                    # (host work beside the running epoch e - 1)
            events[e - 1].synchronize()
            if self._lazy_stats[e - 1, 0].item() == 0 or self._lazy_stats[e - 1, 11].item() != 0:      # that epoch ended the loop (target KL) | exchange timed out
                break
            launch(e)
            executed += 1
        events[-1].synchronize()
        if ev is not None:
            ev[1].record()
        job["ev"] = ev
        # ---- one statistics row as the single launch would have left it
        per = self._lazy_stats[:executed].numpy().astype(np.float32)
        st = np.zeros(32 + E, np.float32)
        stopped = per[-1, 0] == 0
        st[0] = executed - 1 if stopped else E                        # early_stop_epoch
        st[1] = per[:, 1].sum()                                        # optimiser steps
        for k in (2, 3, 4, 5, 6):
            st[k] = np.float32(per[:, k].astype(np.float64).sum())     # sums over the steps: entropy, policy, value, cost-value losses, clip fraction
        st[7:11] = per[-1, 7:11]                                       # mean KL of the last executed epoch, the last minibatch's loss terms
        st[11] = per[:, 11].max()
        st[32:32 + executed] = per[:, 32]
        tail = self.train_readback().cpu().numpy()
        host = np.concatenate([st.astype(np.float64), tail[32 + E:]])
        job["rng_state"] = states + [states[-1]] * (E - len(states))  # (indexable by executed epochs like the single launch's list)
        self._train_end(job, host=host)

    # train() in three pieces (see _rollout_begin): descriptors, the launch (single here, batched in seed_batch.py), read-back + logs
    def _train_begin(self, perms=None, device_perms=None):
        lr = float(self.lr_schedule(self._current_progress_remaining))
        clip_range = float(self.clip_range(self._current_progress_remaining))
        crv = -1.0 if self.clip_range_reward_vf is None else float(self.clip_range_reward_vf(self._current_progress_remaining))
        ccv = -1.0 if self.clip_range_cost_vf is None else float(self.clip_range_cost_vf(self._current_progress_remaining))
        rb, pol, dev = self.rollout_buffer, self.policy, self.device
        n = rb.buffer_size * rb.n_envs
        rng_state, injected = None, perms is not None
        if device_perms is not None:              # (_train_epochwise fills this [n_epochs, n] device tensor epoch by epoch)
            perms = device_perms
        elif perms is None:
            perms, rng_state = self._draw_permutations(n)
        else:
            perms = torch.as_tensor(np.asarray(perms).astype(np.int32), device=dev).contiguous()
        current_penalty = float(self.dual.nu().item())
        if not hasattr(self, "_train_ws"):
            n_words = 96 + 6 * self.n_epochs * (-(-n // int(self.batch_size))) + (self.n_epochs * n + 1) // 2 + 32 + _lib.PPO_SPLIT_BYTES // 8
            self._generic_update = pol.wide or int(self.batch_size) > 256      # shapes of the generic-shape path (csrc/generic.hip)
            if self._generic_update:      # its scratch lies behind the regular workspace: ICRL_PPO_GENERIC_BYTES(batch_size, row_floats, n_params)
                B_, n_ = int(self.batch_size), pol.n_params
                persist = ((B_ + 15) // 16 + 1) * n_ + 1024 if (B_ + 15) // 16 <= 32 and n_ <= 131072 else 0      # ICRL_PPO_GENERIC_PERSIST_FLOATS
                n_words += (64 + B_ * (24 + 1 + 16 + 2 * pol.row_floats) + n_ + (n_ + 255) // 256 + 1088 + persist + 1) // 2 + 8
            # the workspace lives in an arena with room for SYNC_CANDIDATES positions 1 MB apart: see _tune_sync_placement
            arena = torch.zeros(n_words + (self.SYNC_CANDIDATES - 1) * (1 << 17), dtype=torch.int64, device=dev)
            self._train_ws = dict(nu=torch.zeros(1, device=dev), stats=torch.zeros(32 + self.n_epochs, device=dev), sync=arena[:n_words],
                                  sync_arena=arena, sync_words=n_words, sync_tuned=False, t=torch.zeros(1, dtype=torch.int32, device=dev))
        ws = self._train_ws
        ws["nu"].fill_(current_penalty)
        ws["t"].fill_(pol.adam_step)
        hp = PpoHyperT(int(self.batch_size), int(self.n_epochs), int(self.target_kl is not None), int(getattr(self, "profile_phases", 0)) | {"tiles": 2, "rows": 4, "rows1": 12, "pairs": 16, "halves": 32}.get(getattr(self, "train_kernel", "auto"), 0), clip_range, float(self.ent_coef),
                       float(self.reward_vf_coef), float(self.cost_vf_coef), float(self.max_grad_norm),
                       float(self.target_kl or 0.0), crv, ccv, lr, 0.9, 0.999, float(pol.optimizer_kwargs.get("eps", 1e-8)))
        return dict(ps=pol.struct(), bs=rb.struct(), hp=hp, perms=perms, rng_state=rng_state, injected=injected, clip_range=clip_range)

    # The three workgroups of an update exchange one granule per optimiser step through the first 512 bytes of the `sync` workspace, and
    # how fast that hop is depends on where those bytes lie in device memory: moving the workspace by 2 MB switches the step between
    # 8.87-8.90 us and 9.04-9.14 us on MI355X (period 4 MB inside one allocation, sub-structure at 512 KB; across allocations the
    # virtual address does not tell — tools/sync_placement.py).  Which position is the near one is not knowable up front, so the first
    # train() of an agent times a short update (one epoch over <= 16 384 rows of the rollout it is about to train on, identity
    # permutation, parameters and moments restored afterwards) at SYNC_CANDIDATES positions 1 MB apart and keeps the fastest.
    # OFF since the update's workgroups sit on one XCD and store their granules with workgroup scope (csrc/ppo_common.h, "XCD
    # placement"): the hop is served by that XCD's L2 and the position of the bytes in device memory no longer shows — measured with /
    # without the calibration: HC 8.25 / 8.25 us per step, AntWall 19.6 / 19.6.  The mechanism stays for a dispatch that does not give
    # a run one XCD (the kernels then fall back to agent-scope stores): agent.tune_sync_placement = True.
    SYNC_CANDIDATES = 4
    tune_sync_placement = False

    def _tune_sync_placement(self, job):
        ws, pol, rb = self._train_ws, self.policy, self.rollout_buffer
        ws["sync_tuned"] = True
        N, B = rb.n_envs, int(self.batch_size)
        Tc = min(rb.buffer_size, max(1, 16384 // N))
        if not self.tune_sync_placement or Tc * N < 64 * B or getattr(self, "_generic_update", False):       # too few optimiser steps to time a 3 % difference
            return
        from .structs import BufferT, PpoHyperT
        b = _lib.byref
        bs = BufferT.from_buffer_copy(job["bs"]); bs.T = Tc
        hp = PpoHyperT.from_buffer_copy(job["hp"]); hp.n_epochs = 1; hp.use_target_kl = 0; hp._pad = hp._pad & ~1
        perm = torch.arange(Tc * N, dtype=torch.int32, device=self.device)
        keep = [t.clone() for t in (pol.params, pol.exp_avg, pol.exp_avg_sq, ws["t"], ws["stats"])]
        arena, n_words = ws["sync_arena"], ws["sync_words"]
        best = (None, 0)
        for rep_ in range(2):
            for c in range(self.SYNC_CANDIDATES):
                view = arena[c * (1 << 17):c * (1 << 17) + n_words]
                for dst, src in zip((pol.params, pol.exp_avg, pol.exp_avg_sq, ws["t"]), keep):
                    dst.copy_(src)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(_lib.lib().icrl_ppo_lag_train(b(job["ps"]), p(pol.exp_avg), p(pol.exp_avg_sq), p(ws["t"]), b(bs), p(perm), p(ws["nu"]),
                                                         b(hp), p(ws["stats"]), p(view), _lib.current_stream()), "icrl_ppo_lag_train")
                e1.record(); e1.synchronize()
                ms = e0.elapsed_time(e1)
                if rep_ == 1 and (best[0] is None or ms < best[0]):     # (the first round warms the code and the rows up)
                    best = (ms, c)
        for dst, src in zip((pol.params, pol.exp_avg, pol.exp_avg_sq, ws["t"], ws["stats"]), keep):
            dst.copy_(src)
        c = best[1]
        arena.zero_()
        ws["sync"] = arena[c * (1 << 17):c * (1 << 17) + n_words]
        ws["sync_position"] = c

    def _train_launch(self, job):
        pol, ws, b = self.policy, self._train_ws, _lib.byref
        if not ws["sync_tuned"]:
            self._tune_sync_placement(job)
        ev = None
        if getattr(self, "train_events", None) is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _lib.check(_lib.lib().icrl_ppo_lag_train(b(job["ps"]), p(pol.exp_avg), p(pol.exp_avg_sq), p(ws["t"]), b(job["bs"]), p(job["perms"]), p(ws["nu"]),
                                                 b(job["hp"]), p(ws["stats"]), p(ws["sync"]), _lib.current_stream()), "icrl_ppo_lag_train")
        if ev is not None:
            ev[1].record()
        job["ev"] = ev

    def train_readback(self):
        """device tensor with everything _train_end reads back: [32 + n_epochs statistics | adam step | mean / sum of orig_costs |
        mean reward / cost advantages | mean std | rollout status] — stacked over runs by seed_batch.py (one copy for all runs)."""
        rb, pol, ws = self.rollout_buffer, self.policy, self._train_ws
        std = torch.exp(pol.log_std).mean() if pol.log_std is not None else torch.zeros((), device=self.device)
        tail = torch.stack([rb.orig_costs.mean(), rb.orig_costs.sum(), rb.reward_advantages.mean(), rb.cost_advantages.mean(), std])
        return torch.cat([ws["stats"].double(), ws["t"].double(), tail.double(), self._ag["status"].double() + 2.0 * self.rollout_buffer.gae_status.double(),
                          self._explained_variance().double()])

    def _explained_variance(self):
        """device float32 [2]: `explained_variance(returns.flatten(), values.flatten())` of the reward and of the cost critic as the
        reference logs them (ref: ppo_lag.py:311-312 -> common/utils.py:43-59; NB its argument order: y_pred = returns, y_true = values,
        so the figure is 1 - Var[values - returns] / Var[values]) — one pass over the four [T, N] planes (icrl_explained_variance)."""
        rb, ws = self.rollout_buffer, self._train_ws
        if "ev_work" not in ws:
            ws["ev_work"] = torch.zeros(8 * 256, dtype=torch.float64, device=self.device)
            ws["ev_out"] = torch.zeros(2, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().icrl_explained_variance(p(rb.reward_returns), p(rb.reward_values), p(rb.cost_returns), p(rb.cost_values),
                                                      rb.buffer_size * rb.n_envs, p(ws["ev_work"]), p(ws["ev_out"]), _lib.current_stream()),
                   "icrl_explained_variance")
        return ws["ev_out"]

    def _train_end(self, job, host=None):
        """host: this run's row of train_readback() already on the host (numpy float64; float32 values survive the round trip
        exactly), or None: read here."""
        rb, pol, ws = self.rollout_buffer, self.policy, self._train_ws
        rng_state, injected, clip_range, ev = job["rng_state"], job["injected"], job["clip_range"], job.get("ev")
        pol.prepare()                                   # refresh the transposed copy for the next rollout
        self._n_updates += self.n_epochs
        # ---- the scalars the reference logs (one device->host read per train())
        ns = 32 + self.n_epochs
        if host is None:
            average_cost_t = rb.orig_costs.mean()
            total_cost_t = rb.orig_costs.sum()
        st = ws["stats"].cpu().numpy() if host is None else np.asarray(host[:ns], np.float32)
        if st[11] != 0:
            raise RuntimeError("icrl_ppo_lag_train: inter-workgroup exchange timed out")
        if host is None:
            self.check_rollout_status()
        elif int(host[ns + 6]) & 2:
            self.rollout_buffer.check_gae_status()
        elif host[ns + 6] != 0:
            self._ag["status"].zero_()
            raise RuntimeError("icrl_rollout_collect: inter-workgroup exchange timed out (a workgroup of the persistent "
                               "rollout was not resident); rollout buffer and normaliser statistics are invalid")
        pol.adam_step = int(ws["t"].item()) if host is None else int(host[ns])
        steps = max(int(st[1]), 1)
        if ev is not None:
            self.train_events.append((ev[0], ev[1], steps))
        early_stop_epoch = int(st[0])
        if self.streams is not None and not injected:
            self.streams.consumed(min(early_stop_epoch + 1, self.n_epochs))
        if rng_state is not None:       # leave np.random where the reference would: one permutation per executed epoch
            np.random.set_state(rng_state[min(early_stop_epoch + 1, self.n_epochs) - 1])
        if host is None:
            average_cost, total_cost = float(average_cost_t.item()), float(total_cost_t.item())
            mean_ra, mean_ca = float(rb.reward_advantages.mean().item()), float(rb.cost_advantages.mean().item())
            ev_r, ev_c = (float(x) for x in self._explained_variance().cpu().numpy())
        else:
            average_cost, total_cost, mean_ra, mean_ca = (float(x) for x in host[ns + 1:ns + 5])
            ev_r, ev_c = float(host[ns + 7]), float(host[ns + 8])
        if self.update_penalty_after is None or ((self._n_updates / self.n_epochs) % self.update_penalty_after == 0):
            self.dual.update_parameter(np.float32(average_cost))
        logger.record("train/entropy_loss", st[2] / steps)
        logger.record("train/policy_gradient_loss", st[3] / steps)
        logger.record("train/reward_value_loss", st[4] / steps)
        logger.record("train/cost_value_loss", st[5] / steps)
        logger.record("train/approx_kl", float(st[7]))
        logger.record("train/clip_fraction", st[6] / steps)
        logger.record("train/loss", float(st[8] + self.reward_vf_coef * st[9] + self.cost_vf_coef * st[10]))
        logger.record("train/mean_reward_advantages", mean_ra)
        logger.record("train/mean_cost_advantages", mean_ca)
        logger.record("train/reward_explained_variance", ev_r)
        logger.record("train/cost_explained_variance", ev_c)
        logger.record("train/nu", self.dual.nu().item())
        logger.record("train/nu_loss", self.dual.loss.item())
        logger.record("train/average_cost", average_cost)
        logger.record("train/total_cost", total_cost)
        logger.record("train/early_stop_epoch", early_stop_epoch)
        if pol.log_std is not None:
            logger.record("train/std", float(torch.exp(pol.log_std).mean().item()) if host is None else float(host[ns + 5]))
        logger.record("train/n_updates", self._n_updates)
        logger.record("train/clip_range", clip_range)
        logger.record("train/learning_rate", float(self.lr_schedule(self._current_progress_remaining)))      # ref: base_class.py:221
        self.epoch_kls = st[32:32 + self.n_epochs].copy()
        self.phase_cycles = st[12:32].copy()
