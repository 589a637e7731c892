"""The callbacks `cpg` trains with (constraint transfer, BASELINE configs[4]), host side.

ref: stable_baselines3/common/callbacks.py:17-214 (BaseCallback / CallbackList: `n_calls` counts calls of on_step(), i.e.
     VECTORISED env steps, not env-timesteps), :216-247 (CheckpointCallback), :259-379 (EvalCallback),
     icrl/utils.py:542-568 (AdjustedRewardCallback), :607-619 (SaveEnvStatsCallback), icrl/cpg.py:160-198 (how they are wired).

The fused rollout steps all `n_steps` vectorised env steps in one launch, so instead of n_steps on_step() calls a callback
receives ONE on_steps(n): `n_calls` advances by n and every trigger `n_calls % freq == 0` that falls inside the window fires
once, after the launch, with `num_timesteps` set to what the reference would have seen at the triggering call.  The policy
does not change inside a rollout, so for an evaluation this differs from the reference only in the training-env statistics
it syncs (end of the rollout instead of the triggering step); with freq a multiple of n_steps (the default: eval_every =
n_steps = 2048) the trigger IS the rollout's last step and nothing differs.
"""
import os

import numpy as np

from . import logger


class BaseCallback:
    def __init__(self, verbose=0):
        self.model, self.training_env = None, None
        self.n_calls, self.num_timesteps, self.verbose = 0, 0, verbose
        self.locals, self.globals = {}, {}

    def init_callback(self, model):
        self.model, self.training_env = model, model.env
        self._init_callback()

    def _init_callback(self): pass
    def on_training_start(self, locals_=None, globals_=None): self._on_training_start()
    def _on_training_start(self): pass
    def on_rollout_start(self): self._on_rollout_start()
    def _on_rollout_start(self): pass
    def _on_step(self): return True
    def update_locals(self, locals_): self.locals.update(locals_)

    def on_step(self):
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    def on_steps(self, n):
        """n vectorised env steps happened in one launch (see the module docstring)."""
        n_envs = self.model.n_envs
        end_ts = self.model.num_timesteps
        first = self.n_calls + 1
        ok = True
        for call in self._trigger_calls(first, first + n - 1):
            self.n_calls = call
            self.num_timesteps = end_ts - (first + n - 1 - call) * n_envs
            ok = self._on_step() is not False and ok
        self.n_calls = first + n - 1
        self.num_timesteps = end_ts
        return ok

    def _trigger_calls(self, lo, hi):
        """calls in [lo, hi] at which _on_step does something; default: none (a per-step no-op)."""
        return ()

    def on_training_end(self): self._on_training_end()
    def _on_training_end(self): pass
    def on_rollout_end(self): self._on_rollout_end()
    def _on_rollout_end(self): pass


def _multiples(freq, lo, hi):
    freq = int(freq)
    if freq <= 0:
        return ()
    k0 = -(-lo // freq)
    return tuple(range(k0 * freq, hi + 1, freq))


class CallbackList(BaseCallback):
    """ref: callbacks.py:160-214."""

    def __init__(self, callbacks):
        super().__init__()
        self.callbacks = list(callbacks)

    def init_callback(self, model):
        super().init_callback(model)
        for c in self.callbacks:
            c.init_callback(model)

    def on_training_start(self, locals_=None, globals_=None):
        for c in self.callbacks:
            c.on_training_start(locals_, globals_)

    def on_rollout_start(self):
        for c in self.callbacks:
            c.on_rollout_start()

    def on_step(self):
        ok = True
        for c in self.callbacks:
            ok = c.on_step() is not False and ok
        return ok

    def on_steps(self, n):
        ok = True
        for c in self.callbacks:
            ok = c.on_steps(n) is not False and ok
        return ok

    def update_locals(self, locals_):
        for c in self.callbacks:
            c.update_locals(locals_)

    def on_rollout_end(self):
        for c in self.callbacks:
            c.on_rollout_end()

    def on_training_end(self):
        for c in self.callbacks:
            c.on_training_end()


class CheckpointCallback(BaseCallback):
    """ref: callbacks.py:216-247 — model.save(<save_path>/<prefix>_<num_timesteps>_steps) every save_freq calls."""

    def __init__(self, save_freq, save_path, name_prefix="rl_model", verbose=0):
        super().__init__(verbose)
        self.save_freq, self.save_path, self.name_prefix = int(save_freq), save_path, name_prefix
        self.saved = []

    def _init_callback(self):
        if self.save_path is not None:
            os.makedirs(self.save_path, exist_ok=True)

    def _trigger_calls(self, lo, hi):
        return _multiples(self.save_freq, lo, hi)

    def _on_step(self):
        if self.n_calls % self.save_freq == 0 and self.save_path is not None:
            self.saved.append(self.model.save(os.path.join(self.save_path, f"{self.name_prefix}_{self.num_timesteps}_steps")))
        return True


class SaveEnvStatsCallback(BaseCallback):
    """ref: icrl/utils.py:607-619."""

    def __init__(self, env, save_path):
        super().__init__()
        self.env, self.save_path = env, save_path

    def _on_step(self):
        if hasattr(self.env, "obs_rms") and self.save_path:
            self.env.save(os.path.join(self.save_path, "train_env_stats.pkl"))
        return True


class EvalCallback(BaseCallback):
    """ref: callbacks.py:259-379 — every eval_freq calls: sync the normalisation statistics, n_eval_episodes episodes on the
    1-env eval stack, log eval/*, save the best model and fire callback_on_new_best."""

    def __init__(self, eval_env, callback_on_new_best=None, n_eval_episodes=5, eval_freq=10000, best_model_save_path=None,
                 deterministic=True, verbose=1):
        super().__init__(verbose)
        assert eval_env.num_envs == 1, "You must pass only one environment for evaluation"
        self.eval_env, self.callback, self.n_eval_episodes, self.eval_freq = eval_env, callback_on_new_best, n_eval_episodes, int(eval_freq)
        self.best_model_save_path, self.deterministic = best_model_save_path, deterministic
        self.best_mean_reward = self.last_mean_reward = -np.inf
        self.evaluations_timesteps, self.evaluations_results, self.evaluations_length = [], [], []

    def init_callback(self, model):
        super().init_callback(model)
        if self.callback is not None:
            self.callback.init_callback(model)

    def _init_callback(self):
        if self.best_model_save_path is not None:
            os.makedirs(self.best_model_save_path, exist_ok=True)

    def _trigger_calls(self, lo, hi):
        return _multiples(self.eval_freq, lo, hi)

    def _on_step(self):
        if self.eval_freq > 0 and self.n_calls % self.eval_freq == 0:
            from .utils import evaluate_policy
            from .vec_env import sync_envs_normalization
            sync_envs_normalization(self.training_env, self.eval_env)
            rewards, lengths = evaluate_policy(self.model, self.eval_env, n_eval_episodes=self.n_eval_episodes,
                                               deterministic=self.deterministic, return_episode_rewards=True)
            self.evaluations_timesteps.append(self.num_timesteps)
            self.evaluations_results.append(list(rewards)); self.evaluations_length.append(list(lengths))
            mean_reward = float(np.mean(rewards))
            self.last_mean_reward = mean_reward
            logger.record("eval/mean_reward", mean_reward)
            logger.record("eval/mean_ep_length", float(np.mean(lengths)))
            logger.record("eval/best_mean_reward", max(self.best_mean_reward, mean_reward))
            if mean_reward > self.best_mean_reward:
                if self.best_model_save_path is not None:
                    self.model.save(os.path.join(self.best_model_save_path, "best_model"))
                self.best_mean_reward = mean_reward
                if self.callback is not None:
                    self.callback.n_calls, self.callback.num_timesteps = self.n_calls, self.num_timesteps
                    return self.callback._on_step()
        return True


class AdjustedRewardCallback(BaseCallback):
    """ref: icrl/utils.py:542-568 — after every rollout: rollout/adjusted_reward = mean(unnormalised reward - nu * cost) over the
    buffer (the buffer's costs are the NORMALISED ones, as in the reference) and eval/true_cost = mean(cost_fn(orig_obs, actions))."""

    def __init__(self, cost_fn, verbose=1):
        super().__init__(verbose)
        self.cost_fn, self.history = cost_fn, []

    def _on_rollout_end(self):
        rb, env = self.model.rollout_buffer, self.training_env
        rewards = rb.rewards.double()
        if hasattr(env, "ret_rms") and env.norm_reward:          # VecNormalize.unnormalize_reward (vec_normalize.py:130-133)
            rewards = rewards * float(np.sqrt(env.ret_rms.var + env.epsilon))
        adjusted = float((rewards - float(self.model.dual.nu().item()) * rb.costs.double()).mean().item())
        logger.record("rollout/adjusted_reward", adjusted)
        rec = {"adjusted_reward": adjusted}
        if self.cost_fn is not None:
            c = self.cost_fn(rb.orig_observations, rb.actions)
            rec["true_cost"] = float(c.double().mean().item()) if hasattr(c, "double") else float(np.mean(c))
            logger.record("eval/true_cost", rec["true_cost"])
        self.history.append(rec)


class RankSyncCallback(BaseCallback):
    """Multi-GPU cpg (ours; the reference has no multi-device mode): rank r steps its own env shard and runs
    its own PPO epochs; before every rollout but the first, and once more when learn() ends, ONE flat float64 all-reduce
    (icrl_amd/distributed.py: allreduce_state, RCCL over xGMI) averages policy parameters / Adam moments / the dual variable,
    agrees on the step counters and merges the three running-moment sets exactly — the same message as an outer ICRL iteration's,
    without the constraint net (frozen here)."""

    def __init__(self, train_env, world, force_collective=False):
        super().__init__()
        self.train_env, self.world, self.rollouts = train_env, world, 0
        self.force_collective = force_collective       # bench.py's scale anchor: one rank through this path (a 1-rank RCCL group)
        self.sync_events = None                        # a list: (start, end) event pairs around every synchronise()

    def _on_training_start(self):
        from . import distributed as D
        env = self.train_env
        self.rms_list = [r for r in (getattr(env, "obs_rms", None), getattr(env, "ret_rms", None), getattr(env, "cost_rms", None)) if r is not None]
        self.rms_prev = [D._rms_sums(r, "cpu").numpy() for r in self.rms_list]
        self.rollouts = 0

    def synchronise(self):
        from . import distributed as D
        pol, dual = self.model.policy, self.model.dual
        if hasattr(dual, "log_nu"):
            scal = D.Scalars(avg=[(dual, "log_nu"), (dual, "m"), (dual, "v")], counters=[(pol, "adam_step"), (dual, "t")])
        else:
            scal = D.Scalars(avg=[(dual, "pid_i"), (dual, "cost_penalty"), (dual, "_delta_p"), (dual, "_cost_delta")], counters=[(pol, "adam_step")])
        ev = None
        if self.sync_events is not None:
            import torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)); ev[0].record()
        self.rms_prev = D.allreduce_state([pol.params, pol.exp_avg, pol.exp_avg_sq], self.rms_list, self.rms_prev, self.world, scalars=scal,
                                          force_collective=self.force_collective)
        pol.prepare()
        if ev is not None:
            ev[1].record(); self.sync_events.append(ev)

    def _on_rollout_start(self):
        if self.rollouts > 0:
            self.synchronise()
        self.rollouts += 1

    def _on_training_end(self):
        self.synchronise()
