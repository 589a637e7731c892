"""GPU: nominal sampling / evaluation kernel vs the oracle port, and a short end-to-end ICRL run through the reference-
shaped entry point."""
import os

import numpy as np
import pytest
import torch

from oracle import loop as o_loop, nets as o_nets

pytestmark = pytest.mark.gpu


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


def test_sample_from_agent_reference_golden(golden):
    """(s_{t+1}, a_t) pairing, clipped actions, auto-reset observations: the REFERENCE's sample_from_agent output (g9)."""
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    g = golden("g9_learn_iteration")
    train_env = utils.make_train_env("HCWithPos-v0", None, True, 0, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    from icrl_amd.constraint_net import ConstraintNet
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, clip_obs=20, action_low=lo, action_high=-lo)
    train_env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", train_env, n_steps=32, seed=0)
    agent.policy.load_state_dict(_sub(g, "w1/"))
    senv = utils.make_eval_env("HCWithPos-v0", False, seed=0)
    senv.obs_rms.assign(g["obs_rms_mean"], g["obs_rms_var"], float(g["obs_rms_count"]))
    for parallel in (False, True):
        senv.unwrapped.seed(0)
        oo, o, a, r, l = utils.sample_from_agent(agent, senv, 2, noise=g["sample_noise"], parallel=parallel)
        assert list(l) == list(g["sample_lengths"])
        assert np.allclose(oo.cpu().numpy(), g["sample_orig_obs"], rtol=1e-4, atol=2e-5)
        assert np.allclose(o.cpu().numpy(), g["sample_obs"], rtol=1e-4, atol=2e-4)
        assert np.allclose(a.cpu().numpy(), g["sample_actions"], rtol=1e-4, atol=2e-5)
        assert np.allclose(r, g["sample_rewards"], rtol=1e-5, atol=1e-3)


def test_evaluate_policy_wall_termination_vs_port():
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.constraint_net import ConstraintNet
    train_env = utils.make_train_env("HCWithPos-v0", None, True, 3, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, clip_obs=20, action_low=lo, action_high=-lo)
    train_env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", train_env, n_steps=32, seed=3)
    # a policy that drives obs[0] down so that the Test env terminates early
    sd = agent.policy.state_dict()
    B0 = np.random.RandomState(1234).randn(18, 6)[0] * 0.05
    sd["action_net.bias"] = torch.as_tensor(-np.sign(B0) * 2.0, dtype=torch.float32)
    agent.policy.load_state_dict(sd)
    eenv = utils.make_eval_env("HCWithPosTest-v0", False, seed=3)
    rng = np.random.RandomState(0)
    noise = rng.randn(10 * 1000, 6).astype(np.float32) * 0.1
    mean_r, std_r = utils.evaluate_policy(agent, eenv, 10, deterministic=False, noise=noise)
    er, el = utils.evaluate_policy(agent, eenv, 3, deterministic=True, return_episode_rewards=True)
    # oracle port
    stack = o_loop.make_stack(4, "hc", 3)
    port = o_loop.PortAgent(stack, n_steps=32, seed=3)
    port.policy.load_state_dict(sd)
    est = o_loop.make_stack(1, "hc", 3, training=False, norm_reward=False, norm_cost=False, wall_terminate=True)
    port.stack = est
    pm, ps = o_loop.evaluate_policy(port, est, 10, noise)
    assert abs(mean_r - pm) < 1e-3 * max(1, abs(pm)) and abs(std_r - ps) < 1e-3 * max(1, abs(ps))
    assert max(el) < 1000           # the wall was hit


@pytest.mark.parametrize("shape,env_id", [("wide", "HCWithPos-v0"), ("trunk", "HCWithPosTest-v0"), ("deep", "AntWallTest-v0"), ("bare", "HCWithPos-v0")])
def test_generic_shape_sampler_equals_host_loop(shape, env_id):
    """policies of the generic-shape path (layers above 64 units, shared trunk, other depths) in sample_from_agent / evaluate_policy:
    icrl_sample_episodes runs its persistent episode loop with the table-driven forward (sample_episodes_generic_kernel) — parallel
    streams, speculative start rows on the "Test" envs — and must return exactly the rows and episode sums of the reference's loop
    driven from the host over the fine-grained entry points (utils.SteppedEpisodeRun: policy.forward + env.step, same kernels)."""
    from helpers.arches import ARCHES
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    ant = env_id.startswith("Ant")
    od, ad = (113, 8) if ant else (18, 6)
    net_arch = ARCHES.get(shape, [dict(pi=[128, 96], vf=[80, 128], cvf=[128, 128])])
    train_env = utils.make_train_env("AntWall-v0" if ant else "HCWithPos-v0", None, True, 3, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", train_env, n_steps=32, seed=3, policy_kwargs=dict(net_arch=net_arch))
    assert agent.policy.wide
    if env_id.endswith("Test-v0"):      # drive obs[0] down so that episodes end at the wall, at different lengths
        sd = agent.policy.state_dict()
        B0 = np.random.RandomState(1234).randn(od, ad)[0] * 0.05
        sd["action_net.bias"] = torch.as_tensor(-np.sign(B0) * (2.0 if not ant else 1.0), dtype=torch.float32)
        agent.policy.load_state_dict(sd)
    n_ep = 3
    outs = []
    for cls in ("kernel", "host"):
        eenv = utils.make_eval_env(env_id, False, seed=3)
        ms = eenv.unwrapped.max_steps
        noise = np.random.RandomState(0).randn(n_ep * ms, ad).astype(np.float32) * 0.3
        if cls == "kernel":
            run = utils._run_episodes(agent, eenv, n_ep, False, noise, parallel=True)
            assert isinstance(run, utils.EpisodeRun)
        else:
            run = utils.SteppedEpisodeRun(agent, eenv, n_ep, False, noise)
        oo, o, a, r, l = utils.sample_result(run)
        outs.append((oo.cpu().numpy(), o.cpu().numpy(), a.cpu().numpy(), np.asarray(r), np.asarray(l), eenv.unwrapped.step_count.cpu().numpy().copy()))
    k, h = outs
    assert np.array_equal(k[4], h[4]), (k[4], h[4])
    if env_id == "HCWithPosTest-v0":
        assert k[4].min() < 1000
    for i, name in enumerate(("orig_obs", "obs", "actions", "ep_rewards")):
        assert k[i].shape == h[i].shape and np.array_equal(k[i], h[i]), (name, np.abs(k[i] - h[i]).max())
    assert np.array_equal(k[5], h[5])      # the env's random stream is left where the sequential loop leaves it


def test_icrl_entry_point_short_run(tmp_path, golden):
    """`run_me.py icrl`-style flags, 2 outer iterations at a reduced size; checks the metric keys the reference logs."""
    from icrl_amd.icrl import build_parser, icrl
    import types, os
    here = os.path.dirname(os.path.abspath(__file__))
    argv = ["icrl", "-er", "2", "-ep", os.path.join(here, "golden/expert_hc.npz"), "-tk", "0.01", "-cl", "20", "-bi", "4", "-ft", "2000",
            "-ni", "2", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-aclr", "0.9", "-crc", "0.5", "-psis",
            "-ctkno", "2.5", "-nt", "8", "--n_steps", "128", "-s", "0", "--expert_agent_path", os.path.join(here, "golden/expert_hc.npz"),
            "--save_dir", str(tmp_path), "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1)
    metrics, agent, cn, env = icrl(types.SimpleNamespace(**cfg), log=None)
    assert len(metrics) == 2
    for key in ("true/reward", "true/cost", "true/forward_kl", "true/reverse_kl", "forward/nu", "forward/average_cost",
                "forward/approx_kl", "forward/early_stop_epoch", "backward/cn_loss", "backward/kl_new_old", "backward/early_stop_itr",
                "timesteps"):
        assert key in metrics[-1], key
    assert metrics[-1]["timesteps"] == 2 * 2048            # 2 rollouts of 8 x 128 per iteration
    assert all(np.isfinite(m["forward/nu"]) and np.isfinite(m["backward/cn_loss"]) for m in metrics)
    assert os.path.exists(os.path.join(str(tmp_path), "best_cn_model.pt"))
    # the saved constraint net reloads through the (quirky) load path
    from icrl_amd.constraint_net import ConstraintNet
    net = ConstraintNet.load(os.path.join(str(tmp_path), "best_cn_model.pt"))
    assert net.clip_obs is None and net.action_low is None


def test_cpg_transfer_short_run(tmp_path):
    """BASELINE configs[4] shape at reduced size: AntWallBroken with the committed AntBroken constraint net, frozen."""
    import os, types
    from icrl_amd.cpg import build_parser, cpg
    here = os.path.dirname(os.path.abspath(__file__))
    argv = ["cpg", "--cn_path", os.path.join(here, "golden/cn_antbroken.npz"), "-tei", "AntWallBroken-v0", "-eei", "AntWallBrokenTest-v0",
            "-tk", "0.01", "--batch_size", "128", "--reward_gae_lambda", "0.9", "--n_epochs", "3", "--learning_rate", "3e-5",
            "--clip_range", "0.4", "-t", "4096", "-plr", "1.0", "-nt", "16", "--n_steps", "128", "-s", "0", "-v", "0",
            "--save_dir", str(tmp_path), "--eval_every_rollouts", "2"]
    cfg = vars(build_parser().parse_args(argv)); cfg.update(rank=0, world_size=1)
    model, hist = cpg(types.SimpleNamespace(**cfg), log=None)
    assert model.num_timesteps == 4096 and len(hist) == 2 and "eval/mean_reward" in hist[-1] and "eval/mean_reward" not in hist[0]
    assert np.isfinite(hist[-1]["rollout/adjusted_reward"]) and 0.0 <= hist[-1]["eval/true_cost"] <= 1.0
    assert model.env.venv.constraint_net().clip_obs is None          # the load() quirk is what transfer runs evaluate
    assert np.isfinite(model.dual.nu().item())
    from icrl_amd import logger
    assert np.isfinite(logger.Logger.CURRENT.name_to_value["train/average_cost"])


def test_speculative_parallel_evaluation_equals_sequential():
    """evaluate_policy on a "Test" env whose episodes all run to the time limit: the speculative parallel streams give exactly
    the sequential result (rewards, lengths, env left in the same state); see utils._run_episodes."""
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.constraint_net import ConstraintNet
    train_env = utils.make_train_env("HCWithPos-v0", None, True, 5, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, clip_obs=20, action_low=lo, action_high=-lo)
    train_env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", train_env, n_steps=32, seed=5)
    noise = np.random.RandomState(1).randn(4 * 1000, 6).astype(np.float32)
    res = []
    for spec in (True, False):
        utils.SPECULATIVE_EPISODES = spec
        eenv = utils.make_eval_env("HCWithPosTest-v0", False, seed=5)
        r, l = utils.evaluate_policy(agent, eenv, 4, deterministic=False, noise=noise, return_episode_rewards=True)
        res.append((np.array(r), np.array(l), eenv.unwrapped.s.cpu().numpy().copy(), int(eenv.unwrapped.step_count[0].item())))
    utils.SPECULATIVE_EPISODES = True
    assert list(res[0][1]) == [1000] * 4
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]


@pytest.mark.parametrize("n_episodes", [3, 7])
def test_parallel_episodes_with_early_ends_equal_sequential(n_episodes):
    """episodes that END EARLY (the agent is driven into the wall of HCWithPosTest: obs[0] <= -3 after a dozen steps): the parallel
    streams start from guessed positions in the env's random stream / the noise array, and the guess is iterated until it agrees with the
    measured lengths (or, after EpisodeRun.MAX_PASSES passes, the episodes run in one sequential stream) — rewards, lengths,
    recorded rows and the env's final state equal the sequential loop's."""
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.constraint_net import ConstraintNet
    train_env = utils.make_train_env("HCWithPos-v0", None, True, 5, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, clip_obs=20, action_low=lo, action_high=-lo)
    train_env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", train_env, n_steps=32, seed=5)
    B0 = train_env.unwrapped.B.cpu().numpy().reshape(18, 6)[0]
    sd = agent.policy.state_dict()
    sd["action_net.bias"] = torch.as_tensor(-10.0 * np.sign(B0), dtype=torch.float32)      # saturated actions pushing obs[0] down
    agent.policy.load_state_dict(sd)
    noise = np.random.RandomState(2).randn(n_episodes * 1000, 6).astype(np.float32)
    res = []
    for spec in (True, False):
        utils.SPECULATIVE_EPISODES = spec
        eenv = utils.make_eval_env("HCWithPosTest-v0", False, seed=5)
        run = utils._run_episodes(agent, eenv, n_episodes, False, noise, parallel=False)
        res.append((run.out["ep_rewards"].cpu().numpy().copy(), run.lengths.copy(), run.rows_of("orig_obs").cpu().numpy().copy(),
                    run.rows_of("actions").cpu().numpy().copy(), eenv.unwrapped.s.cpu().numpy().copy(), int(eenv.unwrapped.step_count[0].item()),
                    run.passes, run.n_streams))
    utils.SPECULATIVE_EPISODES = True
    assert res[0][1].max() < 200 and len(set(res[0][1].tolist())) > 1            # every episode ended early, at different lengths
    for k in range(6):
        assert np.array_equal(res[0][k], res[1][k]), k
    assert res[0][2].shape[0] == int(res[0][1].sum())
    passes, streams = res[0][6], res[0][7]          # converged as parallel streams after >= 2 passes, or fell back to one stream
    assert (streams == n_episodes and 2 <= passes <= utils.EpisodeRun.MAX_PASSES) or (streams == 1 and passes == utils.EpisodeRun.MAX_PASSES + 1)
    print(f"{n_episodes} early-ending episodes: {passes} passes, {streams} stream(s) in the last one; lengths {res[0][1].tolist()}")


def test_icrl_entry_point_antwall_shapes(tmp_path):
    """BASELINE configs[2] shapes at a reduced size: AntWall-v0 (obs 113, act 8), 2-layer cost net [40, 40], batch 128 (two
    64-row chunks per minibatch in the update kernel), per-step rollout launches with the generic normaliser kernel."""
    from icrl_amd.icrl import build_parser, icrl
    import types, os
    here = os.path.dirname(os.path.abspath(__file__))
    ex = os.path.join(here, "golden/expert_ant.npz")
    argv = ["icrl", "-er", "3", "-ep", ex, "--expert_agent_path", ex, "-tei", "AntWall-v0", "-eei", "AntWallTest-v0", "-tk", "0.02",
            "-cl", "40", "40", "-clr", "0.005", "-crc", "0.6", "-bi", "3", "-ft", "4096", "-ni", "2", "-nt", "32", "--n_steps", "128",
            "-bs", "128", "-ne", "3", "-lr", "3e-5", "-psis", "-ctkno", "2.5", "-s", "0", "--save_dir", str(tmp_path), "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1)
    metrics, agent, cn, env = icrl(types.SimpleNamespace(**cfg), log=None)
    m = metrics[-1]
    assert cn.input_dims == 121 and agent.policy.obs_dim == 113
    for k in ("true/reward", "true/cost", "true/forward_kl", "true/reverse_kl", "forward/nu", "forward/approx_kl", "backward/cn_loss"):
        assert np.isfinite(m[k]), k
    assert m["timesteps"] == 2 * 4096


def test_expert_rollout_files_round_trip(tmp_path):
    """run_policy.save_rollouts writes the reference's per-episode .pkl layout; utils.load_expert_data (which also reads the
    reference's own files/EXPERT/rollouts) reads it back."""
    import pickle
    from icrl_amd import run_policy, utils
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.constraint_net import ConstraintNet
    train_env = utils.make_train_env("HCWithPos-v0", None, True, 2, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, clip_obs=20, action_low=lo, action_high=-lo)
    train_env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", train_env, n_steps=32, seed=2)
    senv = utils.make_eval_env("HCWithPos-v0", False, seed=2)
    root = tmp_path / "expert" / "files" / "EXPERT"
    paths = run_policy.save_rollouts(agent, senv, 3, str(root))
    assert [os.path.basename(p) for p in paths] == ["0.pkl", "1.pkl", "2.pkl"]
    d = pickle.load(open(paths[1], "rb"))
    assert d["observations"].shape == (1000, 18) and d["observations"].dtype == np.float64
    assert d["actions"].shape == (1000, 6) and d["actions"].dtype == np.float32 and np.abs(d["actions"]).max() <= 1.0
    assert d["rewards"].shape == (1,) and d["lengths"].tolist() == [1000] and d["save_scheme"] == "not_airl"
    (obs, acs), mean_reward = utils.load_expert_data(str(tmp_path / "expert"), 3)
    assert obs.shape == (3000, 18) and acs.shape == (3000, 6) and np.array_equal(obs[1000:2000], d["observations"])
    assert np.isfinite(mean_reward)


def test_agent_archive_round_trip(tmp_path):
    """PPOLagrangian.save writes an SB3-style .zip (policy.pth under the reference's names, torch-layout Adam state, dual
    variable); load_parameters restores it, and the archive serves as --expert_agent_path."""
    import zipfile
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.constraint_net import ConstraintNet
    def make(seed):
        env = utils.make_train_env("HCWithPos-v0", None, True, seed, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
        lo = -np.ones(6, np.float32)
        cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, clip_obs=20, action_low=lo, action_high=-lo)
        env.set_cost_function(cn.cost_function)
        return PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=32, batch_size=64, n_epochs=2, seed=seed)
    a = make(1)
    a.learn(2 * 4 * 32)
    path = a.save(str(tmp_path / "best_nominal_model"))
    with zipfile.ZipFile(path) as z:
        assert {"data", "policy.pth", "policy.optimizer.pth", "pytorch_variables.pth"} <= set(z.namelist())
    b = make(9).load_parameters(path)
    for k, v in a.policy.state_dict().items():
        assert torch.equal(v, b.policy.state_dict()[k]), k
    assert torch.equal(a.policy.exp_avg, b.policy.exp_avg) and torch.equal(a.policy.exp_avg_sq, b.policy.exp_avg_sq)
    assert a.policy.adam_step == b.policy.adam_step > 0 and a.dual.nu().item() == b.dual.nu().item()
    # torch's own Adam accepts the optimizer blob (layout check)
    sd = a.policy.state_dict()
    params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    opt = torch.optim.Adam(params, lr=3e-4, eps=1e-5)
    opt.load_state_dict(a.policy.optimizer_state_dict(lr=3e-4))
    expert = utils.load_expert_agent(path)
    obs = torch.randn(5, 18, dtype=torch.float64, device="cuda"); acs = torch.rand(5, 6, device="cuda")
    assert torch.equal(expert.policy.evaluate_actions(obs, acs)[2], a.policy.evaluate_actions(obs, acs)[2])


def test_cpg_pid_lagrangian_and_callback_cadence(tmp_path):
    """cpg --use_pid (icrl/cpg.py:118-158, dual_variable.py:60-122) on HCWithPos shapes: the nu trajectory over 4 rollouts equals
    the CPU port's (same noise / permutation streams; the controller itself is pinned by g11), and the reference's callbacks fire
    at the reference's cadence, counted in VECTORISED env steps (callbacks.py:216-379, icrl/cpg.py:160-176): with n_steps = 64,
    --eval_every 128 evaluates after rollouts 2 and 4, --save_every 96 checkpoints inside rollouts 2 and 3 at calls 96 and 192
    (names carry call x n_envs timesteps), the adjusted reward is logged after every rollout."""
    import os, types
    from icrl_amd.cpg import build_parser, cpg
    from oracle.streams import SeededStreams
    here = os.path.dirname(os.path.abspath(__file__))
    N, T = 4, 64
    argv = ["cpg", "--cn_path", os.path.join(here, "golden/ref_artifacts/antbroken_best_cn_model.pt"), "-tei", "AntWallBroken-v0",
            "-eei", "AntWallBrokenTest-v0", "-tk", "0.01", "--batch_size", "128", "--reward_gae_lambda", "0.9", "--n_epochs", "2",
            "--learning_rate", "3e-5", "--clip_range", "0.4", "-t", str(4 * N * T), "-nt", str(N), "--n_steps", str(T), "-s", "6", "-v", "0",
            "--use_pid", "-kp", "10", "-ki", "0.05", "-kd", "2", "-pidd", "2", "-b", "0.01", "--save_dir", str(tmp_path),
            "--eval_every", "128", "--save_every", "96"]
    cfg = vars(build_parser().parse_args(argv)); cfg.update(rank=0, world_size=1, streams=SeededStreams(31))
    model, hist = cpg(types.SimpleNamespace(**cfg), log=None)
    assert model.num_timesteps == 4 * N * T and len(hist) == 4
    # ---- cadence
    # (round 6: learn() dumps — and thereby clears — the scalar log before every train() like the reference, on_policy_algorithm.py:483-485 /
    # logger.py:492-504: a value logged during rollout 2 is gone by the end of rollout 3)
    assert ["eval/mean_reward" in h for h in hist] == [False, True, False, True]
    assert sorted(os.listdir(tmp_path / "models")) == [f"rl_model_{96 * N}_steps.zip", f"rl_model_{192 * N}_steps.zip"]
    assert os.path.exists(tmp_path / "best_model.zip") and os.path.exists(tmp_path / "train_env_stats.pkl")
    assert all(np.isfinite(h["rollout/adjusted_reward"]) and 0.0 <= h["eval/true_cost"] <= 1.0 for h in hist)
    # ---- nu trajectory vs the CPU port
    pid_kwargs = dict(alpha=0.01, penalty_init=1.0, Kp=10.0, Ki=0.05, Kd=2.0, pid_delay=2, delta_p_ema_alpha=0.5, delta_d_ema_alpha=0.5)
    ocn = o_nets.CostNet(113, 8, [40, 40], False, None, None, None, None, None)
    ocn.load_state_dict(model.env.venv.constraint_net().state_dict())
    stack = o_loop.make_stack(N, "ant", 6, broken=True); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=128, n_epochs=2, learning_rate=3e-5, clip_range=0.4, reward_gae_lambda=0.9,
                            target_kl=0.01, seed=6, pid_kwargs=pid_kwargs)
    nus = []
    streams = SeededStreams(31)
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    for it in range(4):
        port.collect_rollouts(streams.rollout_noise(T, N, 8))
        out = port.train(lambda e: streams.permutation(e, T * N))
        streams.consumed(min(int(out["train/early_stop_epoch"]) + 1, 2))
        nus.append(out["train/nu"])
    # the history records nu when a rollout ends, i.e. BEFORE that rollout's update; train() then moves it
    got = [h["nu"] for h in hist][1:] + [model.dual.nu().item()]
    assert hist[0]["nu"] == 1.0 and np.allclose(got, nus, rtol=0, atol=1e-5), (got, nus)
    assert len(set(np.round(nus, 6))) > 1                     # the controller moved the multiplier



def test_run_policy_entry_point(tmp_path):
    """ref: icrl/run_policy.py — train a few iterations with --save_dir, then `run_policy -ii` loads best_nominal_model.zip +
    train_env_stats.pkl from the run directory and writes per-episode rollout files the expert loader reads back."""
    from icrl_amd import icrl as I, run_policy as RP, utils
    expert = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden/expert_hc.npz")
    save_dir = str(tmp_path / "run")
    I.main(["icrl", "-er", "2", "-ep", expert, "-tk", "0.01", "-cl", "20", "-bi", "2", "-ft", "512", "-ni", "2", "-tei", "HCWithPos-v0",
            "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-crc", "0.5", "-psis", "-nt", "8", "--n_steps", "64", "-ne", "2", "-s", "0", "-v", "0",
            "--save_dir", save_dir])
    assert os.path.exists(os.path.join(save_dir, "best_nominal_model.zip")) and os.path.exists(os.path.join(save_dir, "config.json"))
    assert os.path.exists(os.path.join(save_dir, "models/icrl_0_itrs/nominal_agent.zip"))
    paths = RP.main(["run_policy", "-l", save_dir, "-ii", "-nr", "2", "-e", "HCWithPos-v0"])
    assert [os.path.basename(p) for p in paths] == ["0.pkl", "1.pkl"]
    import pickle
    for pth in paths:                      # the reference's per-episode layout (icrl/run_policy.py:84-98)
        with open(pth, "rb") as fh:
            d = pickle.load(fh)
        assert d["save_scheme"] == "not_airl" and d["observations"].dtype == np.float64 and d["actions"].dtype == np.float32
        assert d["observations"].shape == (1000, 18) and d["actions"].shape == (1000, 6) and int(d["lengths"][0]) == 1000
    # ... and as expert data of a new run: <dir>/files/EXPERT/rollouts/{i}.pkl is what load_expert_data walks (icrl/icrl.py:25-43)
    exp_dir = str(tmp_path / "expert")
    os.makedirs(os.path.join(exp_dir, "files/EXPERT"))
    os.rename(os.path.join(save_dir, "run_policy", "rollouts"), os.path.join(exp_dir, "files/EXPERT/rollouts"))
    (obs, acs), mean_reward = utils.load_expert_data(exp_dir, 2)
    assert obs.shape == (2000, 18) and acs.shape == (2000, 6) and np.isfinite(mean_reward)
    paths = RP.main(["run_policy", "-l", save_dir, "-ii", "-li", "0", "-nr", "1", "-e", "HCWithPos-v0", "-s", "itr0"])
    assert len(paths) == 1


@pytest.mark.parametrize("extra", [["--warmup_timesteps", "256"], ["--reset_policy"], ["--cn_normalize"], ["-nis"],
                                   ["--train_gail_lambda", "-nis"], ["--cn_batch_size", "64"], ["-dno", "-dnr", "-dnc"],
                                   ["-pl", "32", "48", "-rvl", "64", "32", "-cvl", "16", "64"]])
def test_icrl_entry_point_optional_branches(extra):
    """the optional branches of the outer loop the README configurations do not take (icrl/icrl.py:103-117 warm-up with a null cost,
    :201-203 policy reset, constraint-net input normalisation, no importance sampling, the binary-classifier loss, minibatched
    constraint-net updates, no env normalisation, narrower networks): two outer iterations run and log finite metrics."""
    from icrl_amd import icrl as I
    expert = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden/expert_hc.npz")
    argv = ["icrl", "-er", "2", "-ep", expert, "-tk", "0.01", "-cl", "20", "-bi", "3", "-ft", "512", "-ni", "2", "-tei", "HCWithPos-v0",
            "-eei", "HCWithPosTest-v0", "-clr", "0.05", "-crc", "0.5", "-psis", "-nt", "8", "--n_steps", "64", "-ne", "2", "-s", "0", "-v", "0"] + extra
    cfg = vars(I.build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1)
    import types
    metrics, agent, cn, env = I.icrl(types.SimpleNamespace(**cfg), log=None)
    assert len(metrics) == 2
    for m in metrics:
        for k in ("forward/nu", "forward/average_cost", "true/reward", "true/cost", "backward/cn_loss"):
            assert k in m and np.isfinite(float(m[k])), (extra, k, m.get(k))
