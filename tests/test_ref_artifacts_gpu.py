"""GPU: interop with files written by the REFERENCE itself (tests/golden/ref_artifacts: bytes of its committed data files; the
expected values in ref_artifacts_expected.npz were read out of the same files by the reference's own loaders,
oracle/gen_golden.py:fixtures_expert) — SURVEY.md §8 f-2."""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ART = os.path.join(HERE, "golden/ref_artifacts")


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


def test_constraint_net_load_reference_checkpoint(golden):
    """ConstraintNet.load on the reference's best_cn_model.pt (constraint_net.py:323-402 incl. the positional off-by-one):
    costs equal what the reference's own load() + cost_function produced (g2 `antbroken`)."""
    from icrl_amd.constraint_net import ConstraintNet
    g = _sub(golden("g2_cost_function"), "antbroken/")
    cn = ConstraintNet.load(os.path.join(ART, "antbroken_best_cn_model.pt"))
    assert cn.obs_dim == 113 and cn.acs_dim == 8 and list(cn.hidden_sizes) == [40, 40]
    assert cn.clip_obs is None and cn.action_low is None and cn.action_high is None
    assert np.allclose(cn.cost_function(g["obs"], g["acs"]), g["cost"], rtol=2e-5, atol=2e-6)
    for k, v in cn.state_dict().items():
        assert np.array_equal(v.numpy(), g["w/" + k]), k


def test_agent_load_reference_archive(golden):
    """PPOLagrangian.load on the reference's best_model.zip (base_class.py:564-645): weights, Adam state, counters; nu is back at
    penalty_initial_value because _setup_model() re-creates the dual variable after the attributes are restored."""
    from icrl_amd.ppo_lag import PPOLagrangian
    exp = golden("ref_artifacts_expected")
    path = os.path.join(ART, "hc_best_model.zip")
    agent = PPOLagrangian.load(path)
    assert agent.num_timesteps == int(exp["num_timesteps"]) and agent._n_updates == int(exp["n_updates"]) and agent.n_envs == int(exp["n_envs"])
    assert agent.n_steps == 2048 and agent.batch_size == 64 and agent.target_kl == 0.01 and agent.seed == 27
    assert abs(agent.dual.nu().item() - float(exp["nu"])) < 1e-7 and abs(float(exp["nu"]) - 1.0) < 1e-6
    obs = torch.as_tensor(exp["probe_obs"], device="cuda")
    v_r, v_c, lp, ent = agent.policy.evaluate_actions(obs, torch.zeros(7, 6, device="cuda"))
    for got, key in ((v_r, "v_r"), (v_c, "v_c"), (lp, "log_prob"), (ent, "entropy")):
        assert np.allclose(got.cpu().numpy().reshape(-1), exp[key].reshape(-1), rtol=2e-5, atol=2e-5), key
    assert agent.policy.adam_step == int(exp["adam_step"])
    osd = agent.policy.optimizer_state_dict(lr=float(exp["lr"]))
    assert np.array_equal(osd["state"][0]["exp_avg"].numpy(), exp["exp_avg_0"])
    assert np.array_equal(osd["state"][5]["exp_avg_sq"].numpy(), exp["exp_avg_sq_5"])
    assert agent.policy.optimizer_kwargs.get("eps", None) == float(exp["eps"])
    # load_parameters into an existing agent reads the same bytes
    from icrl_amd import utils
    env = utils.make_train_env("HCWithPos-v0", None, True, 0, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    other = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=32, seed=1).load_parameters(path)
    assert torch.equal(other.policy.params, agent.policy.params) and torch.equal(other.policy.exp_avg_sq, agent.policy.exp_avg_sq)
    # and an env of the wrong shape is refused like check_for_correct_spaces does
    ant = utils.make_train_env("AntWall-v0", None, True, 0, 2, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    with pytest.raises(ValueError):
        PPOLagrangian.load(path, env=ant)


def test_vecnormalize_load_reference_statistics(golden):
    from icrl_amd import utils
    from icrl_amd.vec_env import VecNormalizeWithCost
    exp = golden("ref_artifacts_expected")
    env = utils.make_train_env("HCWithPos-v0", None, True, 0, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    vn = VecNormalizeWithCost.load(os.path.join(ART, "hc_train_env_stats.pkl"), env.venv)
    assert np.array_equal(vn.obs_rms.mean, exp["obs_rms_mean"]) and np.array_equal(vn.obs_rms.var, exp["obs_rms_var"])
    assert vn.obs_rms.count == float(exp["obs_rms_count"]) and vn.ret_rms.var == float(exp["ret_rms_var"])
    assert vn.clip_obs == float(exp["clip_obs"]) and [bool(vn.norm_obs), bool(vn.norm_reward)] == [bool(x) for x in exp["norm_flags"][:2]]
    o = vn.normalize_obs(np.zeros((1, 18)))
    want = np.clip((0 - exp["obs_rms_mean"]) / np.sqrt(exp["obs_rms_var"] + 1e-8), -vn.clip_obs, vn.clip_obs)
    assert np.allclose(o.cpu().numpy()[0], want, rtol=1e-12, atol=1e-12)


def test_compute_kl_vs_oracle(golden):
    """utils.compute_kl (icrl/utils.py:421-437, element [1] of evaluate_actions = the cost value: see the docstring) on the
    reference's HC expert data, nominal agent = fresh policy, expert agent = the reference's best_model policy."""
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    from oracle import loop as o_loop, nets as o_nets
    ex = golden("expert_hc")
    env = utils.make_train_env("HCWithPos-v0", None, True, 0, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=32, seed=3)
    expert = utils.load_expert_agent(os.path.join(HERE, "golden/expert_hc.npz"))
    op, oe = o_nets.TwoCriticPolicy(18, 6), o_nets.TwoCriticPolicy(18, 6)
    op.load_state_dict(agent.policy.state_dict()); oe.load_state_dict(_sub(ex, "policy/"))
    obs, acs = ex["observations"][:2000], ex["actions"][:2000]
    d_obs, d_acs = torch.as_tensor(obs, device="cuda"), torch.as_tensor(acs, device="cuda")
    fwd, rev = utils.compute_kl(agent, d_obs, d_acs, expert), utils.compute_kl(expert, d_obs, d_acs, agent)
    o_fwd, o_rev = o_loop.compute_kl(op, obs, acs, oe), o_loop.compute_kl(oe, obs, acs, op)
    assert abs(fwd - o_fwd) < 1e-4 * max(1.0, abs(o_fwd)) and abs(rev - o_rev) < 1e-4 * max(1.0, abs(o_rev))
    assert abs(fwd + rev) < 1e-4 * max(1.0, abs(o_fwd))               # the two are negatives of each other by construction
    one = utils.compute_kl(agent, d_obs, d_acs)
    assert abs(one - o_loop.compute_kl(op, obs, acs)) < 1e-4 * max(1.0, abs(one))


def test_agent_save_load_round_trip_with_narrow_widths(tmp_path):
    """an agent built with -pl 32 16 -rvl 64 32 -cvl 16 64 and non-default hyper-parameters survives save() -> load():
    widths (read off policy.pth), weights, Adam state and the hyper-parameters a continued run needs; continued training works."""
    from icrl_amd import utils
    from icrl_amd.ppo_lag import PPOLagrangian
    arch = [dict(pi=[32, 16], vf=[64, 32], cvf=[16, 64])]
    env = utils.make_train_env("HCWithPos-v0", None, True, 0, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    from icrl_amd.constraint_net import ConstraintNet
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    kw = dict(n_steps=32, batch_size=32, n_epochs=2, learning_rate=1e-4, clip_range=0.3, budget=0.05, penalty_initial_value=0.5,
              penalty_learning_rate=0.02, target_kl=0.03)
    a = PPOLagrangian("TwoCriticsMlpPolicy", env, seed=3, policy_kwargs=dict(net_arch=arch), **kw)
    a.learn(2 * 4 * 32)
    path = a.save(str(tmp_path / "agent"))
    b = PPOLagrangian.load(path, env=env)
    assert b.policy.widths == a.policy.widths == dict(policy_net=(32, 16), value_net=(64, 32), cost_value_net=(16, 64))
    assert torch.equal(b.policy.params, a.policy.params) and torch.equal(b.policy.exp_avg, a.policy.exp_avg)
    assert torch.equal(b.policy.exp_avg_sq, a.policy.exp_avg_sq) and b.policy.adam_step == a.policy.adam_step
    for k, v in kw.items():
        got = getattr(b, k)
        assert (got(1.0) if callable(got) else got) == v, (k, got, v)
    assert b.algo_type == "lagrangian" and b.num_timesteps == a.num_timesteps and b._n_updates == a._n_updates
    assert abs(b.dual.nu().item() - 0.5) < 1e-6            # the reference's nu-reset quirk: back at penalty_initial_value
    b.learn(4 * 32)                                        # continued training runs with the restored widths
    assert np.isfinite(b.policy.params.cpu().numpy()).all()
    # the reference's own archive (64-64 everywhere) still loads through the width inference
    c = PPOLagrangian.load(os.path.join(ART, "hc_best_model.zip"))
    assert c.policy.widths == dict(policy_net=(64, 64), value_net=(64, 64), cost_value_net=(64, 64))


def test_agent_archive_for_the_reference_loader(tmp_path):
    """`PPOLagrangian.save` writes what the REFERENCE's `PPOLagrangian.load` reads (base_class.py:564-645,
    save_util.py:284-418): `data` carries `policy_class`, `observation_space`, `action_space` as pickles by reference to
    stable_baselines3.common.policies.ActorTwoCriticsPolicy / gym.spaces.box.Box, `pytorch_variables.pth` is the reference's empty
    dict, no `.pth` member the reference's loader would take for a state dict is added.  The reference itself does not exist on the
    GPU box: this test writes the archive and this build's deterministic `predict` on 64 observations to gpurun_out/ — committed from
    there as tests/golden/hip_agent_archive.zip / hip_agent_archive_expected.npz — and oracle/verify_agent_archive.py (build
    container; tests/test_oracle_golden.py runs it there) loads that archive with the reference's own loader and compares `predict`."""
    import base64, json, pickle, sys, types, zipfile
    from icrl_amd import utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    env = utils.make_train_env("HCWithPos-v0", None, True, 11, 4, cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
    env.set_cost_function(cn.cost_function)
    a = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=64, batch_size=64, n_epochs=2, seed=11, target_kl=0.01, learning_rate=1e-3)
    a.learn(3 * 4 * 64)
    path = a.save(str(tmp_path / "hip_agent_archive"))
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        assert names == {"data", "policy.pth", "policy.optimizer.pth", "pytorch_variables.pth", "dual_state.json", "_stable_baselines3_version"}
        assert torch.load(io.BytesIO(z.read("pytorch_variables.pth")), weights_only=True) == {}
        data = json.loads(z.read("data"))
    # the pickles name the reference's classes; they decode under any module that offers those names (here: bare stand-ins)
    mods = {}
    for mod, cls_name in (("gym.spaces.box", "Box"), ("gym.spaces.discrete", "Discrete"), ("stable_baselines3.common.policies", "ActorTwoCriticsPolicy")):
        parts = mod.split(".")
        for i in range(1, len(parts) + 1):
            mods.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
        setattr(mods[mod], cls_name, type(cls_name, (), {"__module__": mod}))
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        osp = pickle.loads(base64.b64decode(data["observation_space"][":serialized:"]))
        asp = pickle.loads(base64.b64decode(data["action_space"][":serialized:"]))
        pcl = pickle.loads(base64.b64decode(data["policy_class"][":serialized:"]))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert type(osp).__name__ == "Box" and osp.shape == (18,) and osp.dtype == np.float64 and np.all(np.isinf(osp.low)) and not osp.bounded_below.any()
    assert asp.shape == (6,) and asp.dtype == np.float32 and np.array_equal(asp.low, -np.ones(6, np.float32)) and asp.bounded_above.all()
    assert pcl.__name__ == "ActorTwoCriticsPolicy"
    for k in ("n_steps", "batch_size", "n_epochs", "learning_rate", "clip_range", "target_kl", "policy_kwargs", "n_envs", "seed", "use_sde",
              "algo_type", "penalty_initial_value", "penalty_learning_rate", "budget", "update_penalty_after", "_total_timesteps"):
        assert k in data and not (isinstance(data[k], dict) and ":serialized:" in data[k]), k
    # this build reads its own archive back (spaces through the printable fields, dual variable from dual_state.json)
    b = PPOLagrangian.load(path)
    assert torch.equal(b.policy.params, a.policy.params) and b.policy.adam_step == a.policy.adam_step
    c = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=64, batch_size=64, n_epochs=2, seed=5).load_parameters(path)
    assert c.dual.nu().item() == a.dual.nu().item()
    rng = np.random.RandomState(64)
    obs = rng.randn(64, 18) * 2
    act, _ = a.predict(obs, deterministic=True)
    v_r, v_c, lp, _ = a.policy.evaluate_actions(obs, act)
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    import shutil
    shutil.copy(path, os.path.join(out, "hip_agent_archive.zip"))
    np.savez(os.path.join(out, "hip_agent_archive_expected.npz"), obs=obs, actions=act.cpu().numpy(), v_r=v_r.cpu().numpy().ravel(),
             v_c=v_c.cpu().numpy().ravel(), log_prob=lp.cpu().numpy(), nu=np.float32(a.dual.nu().item()))
    # the committed fixture (written by an earlier run of this test) still loads here
    fix = os.path.join(HERE, "golden", "hip_agent_archive.zip")
    if os.path.exists(fix):
        d = PPOLagrangian.load(fix)
        exp = np.load(os.path.join(HERE, "golden", "hip_agent_archive_expected.npz"))
        got, _ = d.predict(exp["obs"], deterministic=True)
        assert np.allclose(got.cpu().numpy(), exp["actions"], rtol=1e-6, atol=1e-7)


def test_agent_archive_round_trip_above_1000_observation_components(tmp_path):
    """an agent whose observation space has more than 1000 components (numpy elides the printable `low` / `high` fields of the
    archive's `data` entry: '[-3.14 -3.14 ... 3.14]') and bounds that need all 17 digits survives save() -> load(): the exact shape and
    bounds ride in dual_state.json and load() prefers them; the reference-style `data` entry is still written."""
    import json, zipfile
    from icrl_amd import spaces
    from icrl_amd.ppo_lag import PPOLagrangian, _SpacesOnlyEnv
    O = 1024
    low = -np.pi * (1.0 + np.arange(O) / 7.0)
    high = np.e * (1.0 + np.arange(O) / 3.0)
    env = _SpacesOnlyEnv(1, spaces.Box(low, high, (O,), np.float64), spaces.Box(-1.0, 1.0, (3,), np.float32))
    a = PPOLagrangian("TwoCriticsMlpPolicy", env, seed=1, n_steps=8, batch_size=8, policy_kwargs=dict(net_arch=[dict(pi=[128, 128], vf=[128, 128], cvf=[128, 128])]))      # (observations above 128 components: the generic-shape path)
    path = a.save(str(tmp_path / "big"))
    with zipfile.ZipFile(path) as z:
        printable = json.loads(z.read("data"))["observation_space"]["low"]
        assert "..." in printable                                  # (what made the printable field useless)
    b = PPOLagrangian.load(path)
    assert tuple(b.observation_space.shape) == (O,) and np.array_equal(np.asarray(b.observation_space.low, np.float64), low)
    assert np.array_equal(np.asarray(b.observation_space.high, np.float64), high)
    assert torch.equal(b.policy.params, a.policy.params)
    obs = np.random.RandomState(0).randn(3, O)
    a0, _ = a.predict(obs, deterministic=True)
    b0, _ = b.predict(obs, deterministic=True)
    assert torch.equal(torch.as_tensor(a0).cpu(), torch.as_tensor(b0).cpu())
