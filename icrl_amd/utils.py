"""Host helpers of the ICRL loop with the reference's names and argument meaning.

ref: icrl/utils.py:256-303 (make_env / make_train_env / make_eval_env), :323-357 (sample_from_agent),
     :421-437 (compute_kl), :636-655 (get_net_arch); stable_baselines3/common/evaluation.py:10-67 (evaluate_policy);
     icrl/icrl.py:25-43 (load_expert_data); stable_baselines3/common/save_util.py:284-418 (agent zip format).
"""
import io
import os
import pickle
import types
import zipfile

import numpy as np
import torch

from . import _lib, spaces
from .policies import ActorTwoCriticsPolicy
from .structs import EnvT, p
from .vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalize, VecNormalizeWithCost


def make_train_env(env_id, save_dir, use_cost_wrapper, base_seed=0, num_threads=1, normalize_obs=True, normalize_reward=True,
                   normalize_cost=True, env_index_offset=0, device="cuda", **kwargs):
    """ref: icrl/utils.py:265-288.  SubprocVecEnv[num_threads] of gym envs -> one batched device env; env i is seeded
    base_seed + env_index_offset + i (the offset shards envs across GPUs)."""
    # ref: icrl/utils.py:256-263 — make_env() calls set_random_seed(base_seed) (common/utils.py:23-39): python, numpy and torch
    # generators are seeded HERE, so whatever is built next without a seed of its own (the ConstraintNet, icrl.py:88-117) is
    # initialised reproducibly per seed
    import random
    random.seed(base_seed); np.random.seed(base_seed); torch.manual_seed(base_seed)
    env = HipSynthVecEnv.make(env_id, num_threads, base_seed, device, env_index_offset=env_index_offset)
    if use_cost_wrapper:
        env = VecCostWrapper(env)
    if normalize_reward and normalize_cost:
        assert all(key in kwargs for key in ["cost_info_str", "reward_gamma", "cost_gamma"])
        return VecNormalizeWithCost(env, training=True, norm_obs=normalize_obs, norm_reward=normalize_reward, norm_cost=normalize_cost,
                                    cost_info_str=kwargs["cost_info_str"], reward_gamma=kwargs["reward_gamma"], cost_gamma=kwargs["cost_gamma"])
    if normalize_reward:
        return VecNormalizeWithCost(env, training=True, norm_obs=normalize_obs, norm_reward=normalize_reward, norm_cost=normalize_cost,
                                    reward_gamma=kwargs["reward_gamma"])
    return VecNormalizeWithCost(env, training=True, norm_obs=normalize_obs, norm_reward=normalize_reward, norm_cost=normalize_cost)


def make_eval_env(env_id, use_cost_wrapper, normalize_obs=True, seed=0, device="cuda"):
    """ref: icrl/utils.py:290-303 — one env, statistics frozen, rewards / costs not normalised."""
    env = HipSynthVecEnv.make(env_id, 1, seed, device)
    if use_cost_wrapper:
        env = VecCostWrapper(env)
    return VecNormalizeWithCost(env, training=False, norm_obs=normalize_obs, norm_reward=False, norm_cost=False)


def get_net_arch(config):
    """ref: icrl/utils.py:636-655."""
    separate = dict(pi=list(config.policy_layers), vf=list(config.reward_vf_layers), cvf=list(config.cost_vf_layers))
    if getattr(config, "shared_layers", None) is not None:
        return [*config.shared_layers, separate]
    return [separate]


# ---- single-env episode loops, run as ONE persistent kernel launch (icrl_sample_episodes) -----------------------------------
SPECULATIVE_EPISODES = True


class EpisodeRun:
    """`n_episodes` sequential episodes of a 1-env loop (icrl/utils.py:323-357, evaluation.py:10-67) as ONE persistent launch of
    parallel streams, one per episode.  An episode is determined by where it starts in the loop: its start row = the number of
    steps taken before it, which is also its position in the env's random stream and in the action-noise array.  With
    fixed-length episodes the positions are known up front.  When episodes may end early (the "Test" envs, CLGW) the positions
    are GUESSED (every earlier episode runs to the time limit), all episodes run, and the guess is replaced by what the measured
    lengths imply until the two agree: episode 0 is always right, so every pass fixes at least one more episode — in practice
    one or two passes; after MAX_PASSES the episodes run sequentially in one stream.  The rows of a converged pass are exactly
    the sequential loop's.

    In pieces, so that several runs sharing a GPU can put their launches into one grid (icrl_amd/seed_batch.py):
    prepare() builds the descriptors, launch() is the single-run launch, finish(lengths) checks the positions and leaves the env
    where the sequential loop would have left it (returns False when another pass is needed: prepare() again, launch(), ...)."""

    MAX_PASSES = 4

    def __init__(self, agent, env, n_episodes, deterministic, noise, parallel):
        assert env.num_envs == 1, "You must pass only one environment when using this function"
        self.env, self.senv = env, env.unwrapped
        senv = self.senv
        self.pol, self.dev = agent.policy, senv.device
        pol, dev = self.pol, self.dev
        self.n_episodes, self.deterministic = n_episodes, deterministic
        self.max_steps, self.O = senv.max_steps, senv.obs_dim
        self.A = 1 if pol.discrete else pol.act_dim
        self.fixed_len = not senv.wall_terminate
        self.rows = n_episodes * self.max_steps
        if noise is None and not deterministic:
            noise = torch.rand(self.rows, device=dev) if pol.discrete else torch.randn(self.rows, self.A, device=dev)
        if noise is not None:
            noise = torch.as_tensor(noise, device=dev).float().reshape(self.rows, -1).contiguous()
        self.noise = noise
        self.base = None           # position of the env's random stream (read on first use: one small device->host copy)
        was_training = env.training
        env.training = False
        self.nm, self.ps = env.struct(), pol.struct()
        env.training = was_training
        self.lo = self.hi = None
        if not pol.discrete:
            self.lo = torch.as_tensor(pol.action_space.low, device=dev).float().contiguous()
            self.hi = torch.as_tensor(pol.action_space.high, device=dev).float().contiguous()
        # parallel=False keeps fixed-length episodes sequential (a test hook); early-ending envs are tried speculatively unless
        # SPECULATIVE_EPISODES is switched off
        self.n_streams = 1 if (n_episodes == 1 or (self.fixed_len and not parallel) or (not self.fixed_len and not SPECULATIVE_EPISODES)) else n_episodes
        self.starts = np.arange(self.n_streams, dtype=np.int64) * self.max_steps      # start row of every stream (first guess)
        self.passes = 0

    def prepare(self, n_streams=None, base=None):
        senv, dev, O, A = self.senv, self.dev, self.O, self.A
        if n_streams is not None and n_streams != self.n_streams:
            self.n_streams = n_streams
            self.starts = np.arange(n_streams, dtype=np.int64) * self.max_steps
        n_streams = self.n_streams
        if base is not None:
            self.base = int(base) & 0xFFFFFFFF
        if self.base is None:
            self.base = int(senv.step_count[0].item()) & 0xFFFFFFFF
        self.eps_per = self.n_episodes // n_streams
        self.rows_per = self.eps_per * self.max_steps
        # per-stream copies of the env's random-stream position: stream e starts where the sequential loop would be
        sc = ((self.base + self.starts) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        self.st = dict(s=senv.s.repeat(n_streams, 1).contiguous(), t_ep=senv.t_ep.repeat(n_streams).contiguous(),
                       step_count=torch.as_tensor(sc, device=dev), key=senv.key.repeat(n_streams).contiguous())
        st = self.st
        self.e = EnvT(n_streams, O, senv.act_dim, self.max_steps, senv.reward_form, int(senv.wall_terminate), int(senv.broken), 0,
                      p(senv.B), p(st["s"]), p(st["t_ep"]), p(st["step_count"]), p(st["key"]))
        self.row0 = torch.as_tensor(self.starts.astype(np.int32), device=dev) if n_streams > 1 else None
        rows, n_episodes = self.rows, self.n_episodes
        self.out = dict(orig_obs=torch.empty(rows, O, dtype=torch.float64, device=dev), obs=torch.empty(rows, O, dtype=torch.float64, device=dev),
                        actions=torch.empty(rows, A, device=dev), ep_rewards=torch.empty(n_episodes, dtype=torch.float64, device=dev),
                        ep_lengths=torch.empty(n_episodes, dtype=torch.int32, device=dev))
        self.passes += 1
        return self

    def launch(self):
        # streams > 0 of the parallel mode start from a reset as well: in the sequential loop their first state is the
        # auto-reset draw made at exactly this counter value
        b, out = _lib.byref, self.out
        _lib.check(_lib.lib().icrl_sample_episodes(b(self.e), b(self.nm), b(self.ps), p(self.noise), p(self.lo), p(self.hi), self.eps_per, self.rows_per,
                                                   int(self.deterministic), 1, p(self.row0), self.rows, p(out["orig_obs"]), p(out["obs"]),
                                                   p(out["actions"]), p(out["ep_rewards"]), p(out["ep_lengths"]), _lib.current_stream()),
                   "icrl_sample_episodes")

    def finish(self, lengths=None):
        """lengths: ep_lengths already on the host, or None (read here).  False: an episode ended early, so its successors started
        at the wrong position — the positions the measured lengths imply are installed: prepare(), launch() and finish() again."""
        if lengths is None:
            lengths = self.out["ep_lengths"].cpu().numpy()
        lengths = np.asarray(lengths).astype(np.int64)
        if self.n_streams > 1:
            implied = np.concatenate([[0], np.cumsum(lengths)[:-1]])
            if not np.array_equal(implied, self.starts):
                if self.passes >= self.MAX_PASSES:
                    self.n_streams = 1
                    self.starts = np.zeros(1, np.int64)
                else:
                    self.starts = implied
                return False
        senv, env, dev = self.senv, self.env, self.dev
        # leave the env where the sequential loop would have left it
        total = int(lengths.sum())
        senv.step_count.fill_(int(np.uint32((self.base + total) & 0xFFFFFFFF).view(np.int32)))
        senv.s.copy_(self.st["s"][-1:]); senv.t_ep.zero_()
        env.old_obs = senv.s
        self.lengths = lengths
        self.keep = None if total == self.rows else total      # rows are those of the sequential loop: the first `total`
        return True

    def rows_of(self, name):
        x = self.out[name]
        return x if self.keep is None else x[:self.keep]


class SteppedEpisodeRun:
    """EpisodeRun's results from the reference's own loop (icrl/utils.py:323-357, evaluation.py:10-67) over the per-step env chain —
    policy.forward + env.step, one launch sequence and one host read of `done` per step.  Not used by the product path any more (the
    sampler kernels serve every policy shape); tests run it beside EpisodeRun: same rows, same episode sums.  The env is reset before
    the first episode only; the later ones start from the auto-reset observation, like the sequential loop the sampler kernel reproduces."""

    def __init__(self, agent, env, n_episodes, deterministic, noise):
        assert env.num_envs == 1, "You must pass only one environment when using this function"
        pol, senv = agent.policy, env.unwrapped
        dev, max_steps = senv.device, senv.max_steps
        A = 1 if pol.discrete else pol.act_dim
        rows = n_episodes * max_steps
        if noise is None and not deterministic:
            noise = torch.rand(rows, device=dev) if pol.discrete else torch.randn(rows, A, device=dev)
        if noise is not None:
            noise = torch.as_tensor(noise, device=dev).float().reshape(rows, -1).contiguous()
        was_training = env.training
        env.training = False
        try:
            orig, obs_l, acts, ep_rewards, lengths = [], [], [], [], []
            k, obs = 0, env.reset()
            for _ in range(n_episodes):
                done, ep_r, ep_l = False, 0.0, 0
                while not done:
                    actions, _, _, _ = pol.forward(obs, deterministic=deterministic, noise=None if noise is None else noise[k:k + 1])
                    clipped = actions if pol.discrete else pol.last_clipped
                    k += 1
                    obs, r, d, _ = env.step(clipped)
                    orig.append(env.get_original_obs().reshape(1, -1).double().clone()); obs_l.append(obs.reshape(1, -1).double().clone())
                    acts.append(clipped.reshape(1, -1).float().clone())
                    ep_r += float(torch.as_tensor(r).reshape(-1)[0].item()); ep_l += 1
                    done = bool(torch.as_tensor(d).reshape(-1)[0].item())
                ep_rewards.append(ep_r); lengths.append(ep_l)
        finally:
            env.training = was_training
        self.out = dict(orig_obs=torch.cat(orig), obs=torch.cat(obs_l), actions=torch.cat(acts),
                        ep_rewards=torch.as_tensor(np.asarray(ep_rewards, np.float64), device=dev),
                        ep_lengths=torch.as_tensor(np.asarray(lengths, np.int32), device=dev))
        self.lengths, self.keep = np.asarray(lengths, np.int64), None

    def rows_of(self, name):
        return self.out[name]


def _run_episodes(agent, env, n_episodes, deterministic, noise, parallel):
    # (policies of the generic-shape path take the same launch: icrl_sample_episodes runs its persistent loop with the table-driven
    # forward, csrc/rollout.hip sample_episodes_generic_kernel; SteppedEpisodeRun is the same loop from the host, kept as the check)
    run = EpisodeRun(agent, env, n_episodes, deterministic, noise, parallel).prepare()
    run.launch()
    while not run.finish():
        run.prepare().launch()
    return run


def sample_from_agent(agent, env, rollouts, noise=None, parallel=True):
    """ref: icrl/utils.py:323-357.  Returns (orig_observations, observations, actions, rewards, lengths): the first three are
    device tensors [sum(lengths), ...] holding the observation AFTER each step next to the (clipped) action of that step;
    rewards / lengths are numpy arrays per episode.  With parallel=True the `rollouts` fixed-length episodes of the 1-env loop
    run as independent streams whose random-stream counters are offset exactly as the sequential loop would advance them."""
    return sample_result(_run_episodes(agent, env, rollouts, False, noise, parallel))


def sample_result(run, ep_rewards=None):
    return (run.rows_of("orig_obs"), run.rows_of("obs"), run.rows_of("actions"),
            run.out["ep_rewards"].cpu().numpy() if ep_rewards is None else np.asarray(ep_rewards), run.lengths)


def evaluate_policy(model, env, n_eval_episodes=10, deterministic=True, render=False, callback=None, reward_threshold=None,
                    return_episode_rewards=False, noise=None):
    """ref: stable_baselines3/common/evaluation.py:10-67 (sequential episodes on one env)."""
    run = _run_episodes(model, env, n_eval_episodes, deterministic, noise, parallel=False)
    return evaluate_result(run, None, reward_threshold, return_episode_rewards)


def evaluate_result(run, ep_rewards=None, reward_threshold=None, return_episode_rewards=False):
    ep_rewards = run.out["ep_rewards"].cpu().numpy() if ep_rewards is None else np.asarray(ep_rewards)
    if return_episode_rewards:
        return list(ep_rewards), list(run.lengths)
    mean_reward, std_reward = float(np.mean(ep_rewards)), float(np.std(ep_rewards))
    if reward_threshold is not None:
        assert mean_reward > reward_threshold
    return mean_reward, std_reward


def compute_kl(agent_2, observations, actions, agent_1=None):
    """ref: icrl/utils.py:421-437; observations are fed un-normalised, as the reference does.  QUIRK kept: the reference
    takes element [1] of evaluate_actions(), which for the two-critics policy is the cost value (policies.py:752-767 returns
    values, cost_values, log_prob, entropy), so the logged "KL" is mean(V_c^{agent_1} - V_c^{agent_2})."""
    return float(compute_kl_device(agent_2, observations, actions, agent_1).item())


def compute_kl_device(agent_2, observations, actions, agent_1=None):
    """compute_kl without the device->host copy: the float32 scalar stays on the device (callers that log many runs at once fetch
    all of them with one copy, icrl_amd/seed_batch.py)."""
    kl = -agent_2.policy.evaluate_actions(observations, actions)[1].reshape(-1)
    if agent_1 is not None:
        kl = kl + agent_1.policy.evaluate_actions(observations, actions)[1].reshape(-1)
    return kl.sum() / kl.shape[0]


# ---- on-disk artefacts -------------------------------------------------------------------------------------------------------
def load_expert_data(expert_path, num_rollouts):
    """ref: icrl/icrl.py:25-43 — `files/EXPERT/rollouts/{i}.pkl` dicts(observations, actions, rewards, lengths); also accepts the
    re-packed fixtures `<expert_path>.npz` (tests/golden/expert_hc.npz: 500-step rollouts concatenated; expert_lgw.npz carries
    per-rollout `lengths`)."""
    if str(expert_path).endswith(".npz"):
        d = np.load(expert_path)
        rows = int(np.sum(d["lengths"][:num_rollouts])) if "lengths" in d.files else num_rollouts * 500
        mean_reward = float(np.mean(d["rewards"][:num_rollouts])) if "rewards" in d.files else float("nan")
        return (d["observations"][:rows], d["actions"][:rows]), mean_reward
    rewards = []
    obs, acs = [], []
    for i in range(num_rollouts):
        with open(os.path.join(expert_path, "files/EXPERT/rollouts", "%s.pkl" % str(i)), "rb") as f:
            data = pickle.load(f)
        obs.append(data["observations"]); acs.append(data["actions"]); rewards.append(data["rewards"])
    return (np.concatenate(obs, axis=0), np.concatenate(acs, axis=0)), float(np.mean(rewards))


class _RefObject:
    """stand-in for an instance of a reference / gym class inside one of the reference's pickles: keeps the attribute dict."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


class _RefUnpickler(pickle.Unpickler):
    """reads pickles written by the reference (train_env_stats.pkl = a pickled VecNormalizeWithCost, vec_normalize.py:42-64)
    without importing stable_baselines3 / gym: their classes become attribute bags.  Only the exact (module, name) pairs a numpy /
    container payload needs are resolved to real objects — nothing callable with side effects (no builtins.eval / getattr /
    __import__, no numpy.load ...); every other global becomes an inert _RefObject subclass."""

    ALLOWED = {
        ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float64"), ("numpy", "float32"), ("numpy", "int64"), ("numpy", "int32"),
        ("numpy", "bool_"), ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
        ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
        ("collections", "OrderedDict"), ("collections", "deque"), ("collections", "defaultdict"),
        ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"), ("builtins", "complex"), ("builtins", "bytearray"),
        ("builtins", "list"), ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
        ("builtins", "str"), ("builtins", "bytes"), ("builtins", "object"),
        ("copyreg", "_reconstructor"), ("_codecs", "encode"),
    }

    def find_class(self, module, name):
        if (module, name) in self.ALLOWED:
            return super().find_class(module, name)
        return type(name, (_RefObject,), {"__module__": module})


def load_reference_pickle(path):
    with open(path, "rb") as f:
        return _RefUnpickler(f).load()


def parse_sb3_data(raw):
    """the `data` JSON of a stable-baselines3 archive (save_util.py:72-176): plain values as they are; entries the reference
    cloud-pickled keep their printable fields only (enough to rebuild Box / Discrete spaces: dtype, shape, low, high, n)."""
    import json
    d = json.loads(raw) if isinstance(raw, (bytes, str)) else raw
    out = {}
    for k, v in d.items():
        if isinstance(v, dict) and ":serialized:" in v:
            t = v.get(":type:", "")
            if "spaces.box.Box" in t:
                shape = tuple(int(x) for x in (v["shape"] if isinstance(v["shape"], (list, tuple)) else str(v["shape"]).strip("[]()").replace(",", " ").split()))
                num = lambda s_: np.array([float(x) for x in str(s_).strip("[]").replace(",", " ").split()], np.float64).reshape(shape)
                out[k] = spaces.Box(num(v["low"]), num(v["high"]), shape, np.dtype(v["dtype"]).type)
            elif "spaces.discrete.Discrete" in t:
                out[k] = spaces.Discrete(int(v["n"]))
            continue            # functions (lr_schedule, clip_range), buffers, the pickled dual variable: rebuilt by _setup_model
        out[k] = v
    return out


# ---- the WRITER side of the `data` entry: entries the reference cloud-pickles (save_util.py:72-119) ---------------------------------
# cloudpickle.dumps of an importable class, or of an instance of one, is an ordinary pickle that names the class by (module, name):
# the stream below says "gym.spaces.box Box" and carries the instance's __dict__ — no gym is needed to WRITE it, and the reference's
# json_to_data (cloudpickle.loads = pickle.loads) resolves the name in ITS environment.  Protocol 2 (textual GLOBAL opcodes), numpy
# objects under their numpy 1.x module path (numpy 2 still resolves `numpy.core.*`, numpy 1.x has no `numpy._core`).
def _pickle_by_reference(module, name, state=None):
    """pickle of `module.name` itself (state None) or of an instance of it built as cls.__new__(cls) + __dict__.update(state) —
    what object.__reduce_ex__(2) emits for a plain class (copyreg.__newobj__, BUILD)."""
    head = b"\x80\x02c" + module.encode() + b"\n" + name.encode() + b"\n"
    if state is None:
        return head + b"."
    body = pickle.dumps(state, protocol=2)
    assert body[:2] == b"\x80\x02" and body[-1:] == b"."
    body = body[2:-1].replace(b"cnumpy._core.", b"cnumpy.core.")
    return head + b")\x81" + body + b"b."


def _sb3_pickled_entry(type_repr, blob, fields):
    """the JSON object save_util.data_to_json writes for a non-JSON-serialisable item (:type:, :serialized:, printable first-level fields)"""
    import base64
    out = {":type:": type_repr, ":serialized:": base64.b64encode(blob).decode()}
    out.update(fields)
    return out


def sb3_space_entry(space):
    """`observation_space` / `action_space` of an agent archive as the reference stores them: a pickled gym.spaces.box.Box /
    gym.spaces.discrete.Discrete (gym 0.17 attribute set: dtype, shape, low, high, bounded_below, bounded_above, np_random | n, shape,
    dtype, np_random; the generator is stored as None: the load path never samples from a space) + the printable fields."""
    if isinstance(space, spaces.Discrete) or hasattr(space, "n"):
        state = dict(n=int(space.n), shape=(), dtype=np.dtype(np.int64), np_random=None)
        return _sb3_pickled_entry("<class 'gym.spaces.discrete.Discrete'>", _pickle_by_reference("gym.spaces.discrete", "Discrete", state),
                                  dict(n=int(space.n), shape=[], dtype="int64", np_random="None"))
    dt = np.dtype(space.dtype)
    low, high = np.asarray(space.low, dt).copy(), np.asarray(space.high, dt).copy()
    state = dict(dtype=dt, shape=tuple(int(x) for x in space.shape), low=low, high=high,
                 bounded_below=-np.inf < low, bounded_above=np.inf > high, np_random=None)
    return _sb3_pickled_entry("<class 'gym.spaces.box.Box'>", _pickle_by_reference("gym.spaces.box", "Box", state),
                              dict(dtype=str(dt), shape=list(state["shape"]), low=str(low), high=str(high),
                                   bounded_below=str(state["bounded_below"]), bounded_above=str(state["bounded_above"]), np_random="None"))


def sb3_policy_class_entry():
    """`policy_class`: the reference stores the class object itself (pickled by reference to stable_baselines3.common.policies)"""
    return _sb3_pickled_entry("<class 'abc.ABCMeta'>", _pickle_by_reference("stable_baselines3.common.policies", "ActorTwoCriticsPolicy"),
                              {"__module__": "stable_baselines3.common.policies"})


def load_policy_state_dict(path):
    """policy.pth out of a stable-baselines3 agent zip (no gym needed), or the `policy/*` arrays of the .npz fixture."""
    if str(path).endswith(".npz"):
        d = np.load(path)
        return {k[len("policy/"):]: d[k] for k in d.files if k.startswith("policy/")}
    with zipfile.ZipFile(path) as z:
        return torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=False)


def load_expert_agent(path, device="cuda"):
    """stand-in for PPOLagrangian.load(best_model.zip) where only .policy.evaluate_actions is used (icrl/icrl.py:82,251-252)."""
    sd = load_policy_state_dict(path)
    obs_dim = int(np.asarray(sd["mlp_extractor.policy_net.0.weight"]).shape[1])
    act_dim = int(np.asarray(sd["action_net.weight"]).shape[0])
    act_space = spaces.Box(-1.0, 1.0, (act_dim,), np.float32) if "log_std" in sd else spaces.Discrete(act_dim)
    pol = ActorTwoCriticsPolicy(spaces.Box(-np.inf, np.inf, (obs_dim,), np.float64), act_space, device=device)
    pol.load_state_dict(sd)
    return types.SimpleNamespace(policy=pol)
