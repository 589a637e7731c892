"""Several independent ICRL runs (seeds) sharing ONE GPU — inside the launches.

One run occupies 3 CUs during its PPO-Lagrangian update (the three persistent workgroups of icrl_ppo_lag_train) and one
workgroup per environment during a rollout: 1-25 % of an MI355X.  north_star: "independent seeds ... fan out"; the reference runs
seeds as separate processes (README.md:14-21).  Here S runs of one configuration advance in lock-step from ONE host thread and
every phase of an outer iteration is ONE launch (sequence) whose grid carries all runs, run = blockIdx.y:

    rollouts    icrl_rollout_collect_batch   grid (envs, S) persistent workgroups + the dual GAE of all runs, grid (tiles x C, S)
    update      icrl_ppo_lag_train_batch     grid (3, S): 3 S persistent workgroups, each run a chain of dependent optimiser steps
    sampling    icrl_sample_episodes_batch   grid (episodes, S)
    backward    icrl_cn_train_batch          4 launches per constraint-net iteration, grid.y = S
    evaluation  icrl_sample_episodes_batch, the KL metrics per run (small launches)

The host work between the launches is the per-run bookkeeping of the reference's loop (icrl_amd/icrl.py: the same methods, split
into begin / launch / end) plus ONE device->host copy per phase that brings back every run's scalars.  Each run keeps its own
random streams (icrl_amd/streams.py), logger, env stacks, buffers and networks, so it computes exactly what it computes alone
(tests/test_seed_batch_gpu.py: bit-identical parameters, statistics and metrics).

Co-residency: the persistent kernels wait for the other workgroups of the SAME run only.  Workgroups are dispatched x-fastest, so
the runs of a grid become resident oldest first and a run that does not fit yet starts when earlier ones have finished; the
phases are not mixed (no update workgroup ever competes with rollout workgroups for a CU), which is why no admission control
is needed.  Nothing else may occupy the GPU's CUs for long while a batch runs (include/icrl_hip.h: co-residency precondition).
"""
import os
import ctypes
import time

import numpy as np
import torch

from . import _lib, logger, utils
from .streams import PrivateStreams
from .structs import CnTrainJobT, PpoTrainJobT, RolloutJobT, SampleJobT, addr, p
from .true_constraint_net import mean_cost
from .vec_env import sync_envs_normalization


def setup_runs(configs, on_setup=None):
    """icrl.setup() for every config, one after the other (the constructors seed the process-wide generators); each run gets
    its own PrivateStreams unless the config already carries a `streams` object, and its own scalar log."""
    from . import icrl as I
    states = []
    for cfg in configs:
        if getattr(cfg, "streams", None) is None:
            cfg.streams = PrivateStreams(cfg.seed, discrete=cfg.train_env_id in ("LGW-v0", "CLGW-v0"))
        if getattr(cfg, "warmup_timesteps", None) is not None and not hasattr(cfg.streams, "rollout_noise"):
            raise ValueError("seed batch: every run needs its own random streams")
        logger.configure()
        st = I.setup(cfg)
        st["logger"] = logger.Logger.CURRENT
        if on_setup is not None:
            on_setup(st)
        states.append(st)
    torch.cuda.synchronize()
    return states


class _as_run:
    """per-run context of the shared host thread: the run's scalar log is the current one."""

    def __init__(self, st):
        self.st = st

    def __enter__(self):
        self.prev = logger.Logger.CURRENT
        logger.Logger.CURRENT = self.st["logger"]

    def __exit__(self, *exc):
        logger.Logger.CURRENT = self.prev


def _jobs(cls, rows):
    arr = (cls * len(rows))()
    for i, r in enumerate(rows):
        arr[i] = cls(*r)
    return arr


# The placement calibration of a run's exchange workspace (8 short update launches, PPOLagrangian._tune_sync_placement) pays for a run
# that owns the GPU; in a batch the 3 S update workgroups sit on CUs all over the chip and the timed difference disappears — see the
# measurement next to the switch's use in DESIGN.md section 9.  ICRL_SEED_TUNE=1 restores the per-run calibration.
TUNE_PLACEMENT_IN_BATCH = os.environ.get("ICRL_SEED_TUNE", "0") == "1"


class SeedBatch:
    def __init__(self, configs=None, states=None, on_setup=None):
        self.states = setup_runs(configs, on_setup) if states is None else states
        c0 = self.states[0]["config"]
        for st in self.states[1:]:
            c = st["config"]
            # everything a batched launch takes from run 0 only: the grids' shapes, and the discounting of the batched GAE
            # (icrl_rollout_collect_batch takes the four gamma / lambda values as scalars)
            same = ("train_env_id", "eval_env_id", "num_threads", "n_steps", "batch_size", "n_epochs", "forward_timesteps", "expert_rollouts",
                    "backward_iters", "cn_batch_size", "cn_layers", "n_iters", "reward_gamma", "reward_gae_lambda", "cost_gamma", "cost_gae_lambda")
            diff = [k for k in same if getattr(c, k) != getattr(c0, k)]
            if diff:
                raise ValueError(f"seed batch: the runs of a batch share every grid; they differ in {diff}")
        if any(st["agent"].policy.wide or int(st["agent"].batch_size) > 256 or getattr(st["constraint_net"], "wide", False) for st in self.states):
            raise ValueError("seed batch: hidden widths above 64 / batch sizes above 256 run on the generic-shape path, which has no batched form")
        if any(st.get("world", 1) > 1 for st in self.states):
            # outer_iteration() below has no per-iteration all-reduce and no rank-0 guard on the saves: a seed batch is a one-rank matter
            raise ValueError("seed batch: the runs of a batch live on ONE rank (launch the batch per GPU; world_size > 1 states would skip the collective)")
        S = len(self.states)
        self.dev = self.states[0]["agent"].device
        self.args_ws = torch.empty(2 * S * _lib.BATCH_ARGS_BYTES, dtype=torch.uint8, device=self.dev)

    # ---- one device->host copy for all runs --------------------------------------------------------------------------------
    @staticmethod
    def _to_host(rows):
        return torch.stack([r.reshape(-1) for r in rows]).cpu().numpy()

    # ---- the batched launches ------------------------------------------------------------------------------------------------
    def _ws(self):
        return p(self.args_ws), self.args_ws.numel()

    def _launch_rollouts(self, agents, jobs):
        a0 = agents[0]
        arr = _jobs(RolloutJobT, [(addr(j["env"]), addr(j["nm"]), addr(j["pol"]), addr(j["cn"]), addr(j["buf"]), addr(j["ag"]), p(j["noise"])) for j in jobs])
        L = _lib.lib()
        ws, nbytes = self._ws()
        err = L.icrl_rollout_collect_batch(len(jobs), arr, p(a0._alow), p(a0._ahigh), float(a0.reward_gamma), float(a0.reward_gae_lambda),
                                           float(a0.cost_gamma), float(a0.cost_gae_lambda), 1, ws, nbytes, _lib.current_stream())
        if err == 1 and b"batched form" in L.icrl_last_error():       # shapes without a batched persistent kernel: one launch per run
            L.icrl_clear_error()
            for a, j in zip(agents, jobs):
                a._rollout_launch(j)
            return
        _lib.check(err, "icrl_rollout_collect_batch")

    def _launch_trains(self, agents, jobs):
        rows = []
        for a, j in zip(agents, jobs):
            ws = a._train_ws
            if not ws["sync_tuned"]:          # first update of the run: where its exchange workspace is fastest (PPOLagrangian._tune_sync_placement)
                if not TUNE_PLACEMENT_IN_BATCH and len(jobs) > 1:
                    a.tune_sync_placement = False
                a._tune_sync_placement(j)
            rows.append((addr(j["ps"]), p(a.policy.exp_avg), p(a.policy.exp_avg_sq), p(ws["t"]), addr(j["bs"]), p(j["perms"]), p(ws["nu"]), addr(j["hp"]),
                         p(ws["stats"]), p(ws["sync"])))
        arr = _jobs(PpoTrainJobT, rows)
        ws, nbytes = self._ws()
        _lib.check(_lib.lib().icrl_ppo_lag_train_batch(len(jobs), arr, ws, nbytes, _lib.current_stream()), "icrl_ppo_lag_train_batch")

    def _launch_episodes(self, runs):
        r0 = runs[0]
        arr = _jobs(SampleJobT, [(addr(r.e), addr(r.nm), addr(r.ps), p(r.noise), p(r.row0), r.rows, 0, p(r.out["orig_obs"]), p(r.out["obs"]),
                                  p(r.out["actions"]), p(r.out["ep_rewards"]), p(r.out["ep_lengths"])) for r in runs])
        ws, nbytes = self._ws()
        _lib.check(_lib.lib().icrl_sample_episodes_batch(len(runs), arr, p(r0.lo), p(r0.hi), r0.eps_per, r0.rows_per, int(r0.deterministic), 1,
                                                         ws, nbytes, _lib.current_stream()), "icrl_sample_episodes_batch")

    def _launch_cn_trains(self, nets, jobs):
        rows = []
        for cn, j in zip(nets, jobs):
            rows.append((addr(j["s"]), p(cn.exp_avg), p(cn.exp_avg_sq), p(j["t_dev"]), p(j["nominal"]), p(j["expert"]), j["nominal"].shape[0],
                         j["expert"].shape[0], p(j["d_off"]), p(j["d_rowep"]), j["n_ep"], 0, addr(j["hp"]), p(j["work"]), p(j["metrics"])))
        arr = _jobs(CnTrainJobT, rows)
        ws, nbytes = self._ws()
        _lib.check(_lib.lib().icrl_cn_train_batch(len(jobs), arr, ws, nbytes, _lib.current_stream()), "icrl_cn_train_batch")

    def _prefetch_permutations(self, agents, after=None):
        """`after`: an event recorded on the main stream BEFORE the update launch — the side stream waits for that point only, so the
        sorts run beside the update kernels (3 S busy CUs) instead of behind them.  (The draws' order is the host's call order: the
        Philox offset of the generator advances on the host.)"""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        if after is not None:
            self._side.wait_event(after)
        else:
            self._side.wait_stream(torch.cuda.current_stream())
        for a in agents:
            s_ = a.streams
            if s_ is not None and hasattr(s_, "prefetch_permutations"):
                s_.prefetch_permutations(a.n_epochs, a.rollout_buffer.buffer_size * a.rollout_buffer.n_envs, self._side)

    # ---- phases ----------------------------------------------------------------------------------------------------------------
    def _learn(self, total_timesteps):
        """PPOLagrangian.learn(total_timesteps, cost_function="cost") of every run (ref: on_policy_algorithm.py:430-492), rollouts
        and updates in lock-step."""
        sts = self.states
        agents = [st["agent"] for st in sts]
        totals = []
        for st, a in zip(sts, agents):
            with _as_run(st):
                totals.append(a._setup_learn(total_timesteps, True))
                if not a._fused_rollout_ok("cost", a.n_steps, a.rollout_buffer):
                    raise ValueError("seed batch: the env chain must be the device-native stack with a ConstraintNet cost (the fused rollout)")
        iteration = 0
        while agents[0].num_timesteps < totals[0]:
            jobs = []
            for st, a in zip(sts, agents):
                with _as_run(st):
                    jobs.append(a._rollout_begin(None, a.rollout_buffer, a.n_steps, None, zero_buffer=False))
            self._launch_rollouts(agents, jobs)
            iteration += 1
            tjobs = []
            for st, a, j in zip(sts, agents, jobs):
                with _as_run(st):
                    a._rollout_end(j, a.env, None, a.rollout_buffer, a.n_steps)
                    a._current_progress_remaining = 1.0 - float(a.num_timesteps) / float(totals[0])
                    a._training_infos(iteration)
                    logger.dump(step=a.num_timesteps)
                    tjobs.append(a._train_begin(None))
            before_update = torch.cuda.Event()
            before_update.record()
            self._launch_trains(agents, tjobs)
            # while the updates run (3 S CUs busy, the host idle): the permutations of the NEXT update, on a side stream — their sorts
            # were ~5 ms of device time + ~7 ms of launches per update phase at S = 32, in front of the update launch.  Not across the end
            # of the forward step when the constraint net draws minibatch permutations from the same generator in between.
            more = agents[0].num_timesteps < totals[0]
            if more or sts[0]["constraint_net"].batch_size is None:
                self._prefetch_permutations(agents, after=before_update)
            host = self._to_host([a.train_readback() for a in agents])          # waits for the update of every run
            for st, a, j, h in zip(sts, agents, tjobs, host):
                with _as_run(st):
                    a._train_end(j, host=h)
        for st, a in zip(sts, agents):
            with _as_run(st):
                a._training_infos(iteration + 1)

    def _episodes(self, envs, n_episodes, deterministic, noises, parallel):
        """n_episodes of the 1-env loop of every run (utils.EpisodeRun) in one launch; runs whose speculation failed repeat
        sequentially (one more batched launch for those)."""
        sts = self.states
        runs = [utils.EpisodeRun(st["agent"], env, n_episodes, deterministic, noise, parallel) for st, env, noise in zip(sts, envs, noises)]
        bases = self._to_host([r.senv.step_count[:1].to(torch.int64) for r in runs])[:, 0]
        for r, b in zip(runs, bases):
            r.prepare(base=int(b))
        todo = list(range(len(runs)))
        rewards = [None] * len(runs)
        dbg = os.environ.get("ICRL_SEED_DEBUG")
        while todo:
            if dbg:
                torch.cuda.synchronize(); t_dbg = time.time()
            sub = [runs[i] for i in todo]
            if len({r.n_streams for r in sub}) == 1:
                self._launch_episodes(sub)
            else:
                for r in sub:
                    r.launch()
            host = self._to_host([torch.cat([r.out["ep_lengths"].double(), r.out["ep_rewards"]]) for r in sub])
            again = []
            for i, r, h in zip(todo, sub, host):
                if r.finish(h[:n_episodes]):
                    rewards[i] = h[n_episodes:].copy()
                else:           # an episode ended early: the run repeats with the positions the measured lengths imply
                    r.prepare()
                    again.append(i)
            if dbg:
                torch.cuda.synchronize()
                print(f"[episodes] pass over {len(todo)} run(s), streams {sorted({r.n_streams for r in sub})}: {1e3 * (time.time() - t_dbg):.1f} ms, {len(again)} to repeat", flush=True)
            todo = again
        return runs, rewards

    def outer_iteration(self, itr):
        """icrl.outer_iteration (ref: icrl/icrl.py:199-304) for every run: the same calls in the same order per run, each phase
        issued for all runs at once."""
        import os
        sts = self.states
        cfg0 = sts[0]["config"]
        for st in sts:
            if st["config"].reset_policy and itr != 0:
                with _as_run(st):
                    old_agent = st["agent"]
                    st["agent"] = st["create_nominal_agent"]()
                    if hasattr(old_agent, "_train_ws"):
                        st["agent"]._train_ws = old_agent._train_ws
        agents = [st["agent"] for st in sts]
        progress = 1 - float(itr) / float(cfg0.n_iters)
        # ---- forward step
        self._learn(cfg0.forward_timesteps)
        fwd = []
        for st, a in zip(sts, agents):
            fwd.append(dict(st["logger"].name_to_value))
            st["timesteps"] += a.num_timesteps
        # ---- nominal trajectories
        A = 1 if agents[0].policy.discrete else agents[0].policy.act_dim
        noises = []
        for st in sts:
            sync_envs_normalization(st["train_env"], st["sampling_env"])
            s_ = getattr(st["config"], "streams", None)
            noises.append(None if s_ is None else s_.sample_noise(cfg0.expert_rollouts * st["sampling_env"].unwrapped.max_steps, A))
        runs, rewards = self._episodes([st["sampling_env"] for st in sts], cfg0.expert_rollouts, False, noises, True)
        samples = [utils.sample_result(r, rw) for r, rw in zip(runs, rewards)]
        # ---- backward step
        nets = [st["constraint_net"] for st in sts]
        cjobs = []
        for st, cn, (orig_obs, obs, acts, rews, lengths) in zip(sts, nets, samples):
            mean, var = None, None
            if st["config"].cn_normalize:
                mean, var = st["sampling_env"].obs_rms.mean, st["sampling_env"].obs_rms.var
            perms = None
            s_ = getattr(st["config"], "streams", None)
            if cn.batch_size is not None and s_ is not None and hasattr(s_, "cn_permutations"):
                perms = s_.cn_permutations(cfg0.backward_iters, min(int(orig_obs.shape[0]), int(np.asarray(cn.expert_obs).shape[0])))
            cjobs.append(cn._train_begin(cfg0.backward_iters, orig_obs, acts, lengths, mean, var, progress, perms))
        if nets[0].batch_size is None:
            self._launch_cn_trains(nets, cjobs)
        else:
            for cn, j in zip(nets, cjobs):
                cn._train_launch(j)
        host = self._to_host([torch.cat([j["metrics"].reshape(-1).double(), j["t_dev"].double()]) for j in cjobs])
        backward = []
        for st, cn, j, h in zip(sts, nets, cjobs, host):
            backward.append(cn._train_end(j, metrics_host=h[:-1].reshape(j["metrics"].shape), adam_step_host=h[-1]))
            st["train_env"].set_cost_function(cn.cost_function)
        # ---- evaluation
        tail = []
        for st, (orig_obs, obs, acts, rews, lengths) in zip(sts, samples):
            tail.append(torch.stack([(orig_obs[..., 0] < -3).double().mean(), (orig_obs[..., 0] > 3).double().mean()]))
            sync_envs_normalization(st["train_env"], st["eval_env"])
        enoise = []
        for st in sts:
            s_ = getattr(st["config"], "streams", None)
            enoise.append(None if s_ is None else s_.eval_noise(10 * st["eval_env"].unwrapped.max_steps, A))
        eruns, erewards = self._episodes([st["eval_env"] for st in sts], 10, False, enoise, False)
        for i, (st, a) in enumerate(zip(sts, agents)):          # the KL metrics stay on the device until the one copy below
            if st["expert_agent"] is not None:
                orig_obs, obs, acts, rews, lengths = samples[i]
                fkl = utils.compute_kl_device(a, st["d_expert_obs"], st["d_expert_acs"], st["expert_agent"])
                rkl = utils.compute_kl_device(st["expert_agent"], orig_obs, acts, a)
                tail[i] = torch.cat([tail[i], torch.stack([fkl, rkl]).double()])
        tail_h = self._to_host(tail)
        out = []
        for i, (st, a) in enumerate(zip(sts, agents)):
            config = st["config"]
            orig_obs, obs, acts, rews, lengths = samples[i]
            average_true_cost = mean_cost(st["true_cost_function"], orig_obs, acts)
            samples_behind, samples_infront = float(tail_h[i][0]), float(tail_h[i][1])
            average_true_reward, std_true_reward = utils.evaluate_result(eruns[i], erewards[i])
            forward_kl = reverse_kl = float("nan")
            if st["expert_agent"] is not None:
                forward_kl, reverse_kl = float(tail_h[i][2]), float(tail_h[i][3])
            best = st["best"]
            if config.save_dir and itr % config.save_every == 0:
                path = os.path.join(config.save_dir, f"models/icrl_{itr}_itrs")
                os.makedirs(path, exist_ok=True)
                a.save(os.path.join(path, "nominal_agent"))
                torch.save(a.policy.state_dict(), os.path.join(path, "nominal_agent_policy.pth"))
                st["constraint_net"].save(os.path.join(path, "cn.pt"))
                st["train_env"].save(os.path.join(path, f"{itr}_train_env_stats.pkl"))
            if average_true_reward > best["reward"] and config.save_dir:
                a.save(os.path.join(config.save_dir, "best_nominal_model"))
                torch.save(a.policy.state_dict(), os.path.join(config.save_dir, "best_nominal_model_policy.pth"))
                st["constraint_net"].save(os.path.join(config.save_dir, "best_cn_model.pt"))
                st["train_env"].save(os.path.join(config.save_dir, "train_env_stats.pkl"))
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
            metrics.update({k.replace("train/", "forward/"): v for k, v in fwd[i].items()})
            metrics.update(backward[i])
            out.append(metrics)
        return out

    def run(self, first_iteration, n_iters):
        """outer iterations [first_iteration, first_iteration + n_iters) of every run.  Returns (metrics per run, wall seconds)."""
        out = [[] for _ in self.states]
        torch.cuda.synchronize()
        t0 = time.time()
        for it in range(first_iteration, first_iteration + n_iters):
            for i, m in enumerate(self.outer_iteration(it)):
                out[i].append(m)
        torch.cuda.synchronize()
        return out, time.time() - t0


def run_iterations(states, first_iteration, n_iters):
    """outer iterations of every run of `states` (setup_runs), all runs in every launch.  Returns (metrics per run, wall seconds)."""
    return SeedBatch(states=states).run(first_iteration, n_iters)


def run_seed_batch(configs, n_iters, on_setup=None):
    """configs: one icrl config (types.SimpleNamespace, see icrl.build_parser) per run.  Runs `n_iters` outer iterations of
    every run.  Returns (states, metrics per run, wall seconds of the iteration phase)."""
    sb = SeedBatch(configs, on_setup=on_setup)
    out, dt = sb.run(0, n_iters)
    return sb.states, out, dt
