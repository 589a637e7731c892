"""GPU parity for BASELINE configs[0] (LapGridWorld, Discrete(2) actions, Categorical policy, one-hot cost net, no
normalisation): the HIP env against the reference's own env traces, the fused rollout + persistent update kernel against
the reference's learn() (golden g10, teacher-forced actions / permutations), the Categorical branch of the update kernel
against the oracle at a wider class count, and the entry point with the README's flags at a reduced size.

Tolerances: env traces are exact; buffers / parameters carry fp32 rounding of tanh, MFMA dot products and Adam
(documented per assert)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import loop as o_loop, nets as o_nets, ppo as o_ppo

pytestmark = pytest.mark.gpu
ADAM_DEV_BOUND = 5e-4      # 3 x the largest deviation measured on MI355X (1.7e-4 x lr x steps; absolute: <= 9e-8, ~1 ulp)
HERE = os.path.dirname(os.path.abspath(__file__))


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


@pytest.mark.parametrize("env_id,key", [("LGW-v0", "lgw"), ("CLGW-v0", "clgw")])
def test_env_matches_reference_trace(golden, env_id, key):
    from icrl_amd.vec_env import HipSynthVecEnv
    g = golden("g10_lap_grid")
    acts = g[key + "/actions"]
    env = HipSynthVecEnv.make(env_id, acts.shape[1])
    assert env.action_space.n == 2 and env.observation_space.shape == (1,)
    obs = env.reset()
    assert np.array_equal(obs.cpu().numpy(), -np.ones((acts.shape[1], 1)))
    for t in range(acts.shape[0]):
        o, r, d, _ = env.step(torch.as_tensor(acts[t], device="cuda"))
        assert np.array_equal(o.cpu().numpy(), g[key + "/obs"][t]), t
        assert np.array_equal(r.cpu().numpy(), g[key + "/rew"][t]), t
        assert np.array_equal(d.cpu().numpy().astype(bool), g[key + "/done"][t]), t


def _lgw_agent(N, T, seed, **kw):
    from icrl_amd import utils
    from icrl_amd.constraint_net import ConstraintNet
    from icrl_amd.ppo_lag import PPOLagrangian
    env = utils.make_train_env("LGW-v0", None, True, seed, N, normalize_obs=False, normalize_reward=False, normalize_cost=False)
    cn = ConstraintNet(1, 2, [20], None, lambda x: 0.003, None, None, True, clip_obs=20)
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=seed, **kw)
    return agent, env, cn


def test_learn_matches_reference_golden(golden):
    """two rollouts + two updates of the REFERENCE (N=2, T=64, batch 16, 3 epochs, ent_coef 0.01) on LGW-v0."""
    g = golden("g10_lap_grid")
    acts = g["learn_actions"]
    _, T, N = acts.shape
    agent, env, cn = _lgw_agent(N, T, 3, batch_size=16, n_epochs=3, target_kl=0.01, penalty_initial_value=1, penalty_learning_rate=0.1,
                                ent_coef=0.01)
    cn.load_state_dict(_sub(g, "cn/"))
    agent.policy.load_state_dict(_sub(g, "w0/"))
    perms = [g["perms0"][:g["n_perms"][0]], g["perms1"][:g["n_perms"][1]]]
    agent._setup_learn(2 * N * T)
    for it in range(2):
        noise = torch.as_tensor(acts[it].astype(np.float32), device="cuda")     # uniform 0 / 1 forces class 0 / 1
        agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=noise)
        if it == 1:
            rb = agent.rollout_buffer
            # exact fields: integer / table-lookup arithmetic
            for k in ("actions", "rewards", "dones", "orig_observations", "observations", "new_observations"):
                assert np.array_equal(getattr(rb, k).cpu().numpy().reshape(T, N, -1), g["buf/" + k]), k
            # network outputs after one update on each side: fp32 rounding of the update (lr 3e-4) and of tanh / exp
            for k in ("log_probs", "reward_values", "cost_values", "costs", "orig_costs", "reward_advantages", "cost_advantages",
                      "reward_returns", "cost_returns"):
                got, ref = getattr(rb, k).cpu().numpy().reshape(T, N, -1), g["buf/" + k]
                assert np.allclose(got, ref, rtol=1e-3, atol=2e-4), (k, np.abs(got - ref).max())
        p = np.zeros((3, N * T), np.int64)
        p[:len(perms[it])] = perms[it]
        agent.train(perms=p)
    from icrl_amd import logger
    lg = logger.Logger.CURRENT.name_to_value
    assert lg["train/early_stop_epoch"] == g["log/early_stop_epoch"].item()
    for k, v in agent.policy.state_dict().items():
        ref = g["w1/" + k]
        dev = float(np.abs(v.numpy() - ref).max())
        print(f"ADAM_DEV lgw-golden {k}: {dev:.3g} = {dev / (3e-4 * 48):.3g} x lr x steps")
        assert dev <= ADAM_DEV_BOUND * 3e-4 * 48 + 2e-7, (k, dev)
    assert abs(lg["train/nu"] - g["log/nu"].item()) < 1e-5
    assert abs(lg["train/entropy_loss"] - g["log/entropy_loss"].item()) < 1e-4
    assert abs(lg["train/policy_gradient_loss"] - g["log/policy_gradient_loss"].item()) < 2e-4
    assert abs(lg["train/approx_kl"] - g["log/approx_kl"].item()) < 2e-5


@pytest.mark.parametrize("O,A,N,T,B,E,ent,wide", [(18, 5, 8, 32, 64, 3, 0.01, False), (1, 2, 4, 40, 64, 2, 0.0, False), (40, 16, 8, 16, 128, 2, 0.05, False),
                                                  # the Categorical branch of the generic-shape path (csrc/generic.hip): layers above 64 / a 320-row batch
                                                  (18, 5, 8, 32, 64, 3, 0.01, True), (1, 2, 8, 80, 320, 2, 0.02, False),
                                                  # ... and of a policy with a shared trunk / other depths (icrl_policy_t.arch)
                                                  (18, 5, 8, 32, 64, 3, 0.01, "trunk"), (7, 3, 4, 50, 100, 2, 0.03, "deep")])
def test_categorical_update_vs_oracle(O, A, N, T, B, E, ent, wide):
    """the Categorical branch of the update kernel at other widths (up to 16 classes) against the oracle epoch loop."""
    from helpers.arches import ARCHES, oracle_arch_kwargs
    from icrl_amd import logger, spaces
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    rng = np.random.RandomState(O * A + T)
    senv = HipSynthVecEnv(N, "hc", 0)
    senv.observation_space = spaces.Box(-np.inf, np.inf, (O,), np.float64)
    senv.action_space = spaces.Discrete(A)
    env = VecNormalizeWithCost(VecCostWrapper(senv))
    net_arch = ARCHES[wide] if isinstance(wide, str) else [dict(pi=[128, 72], vf=[100, 128], cvf=[128, 128])]
    akw = dict(policy_kwargs=dict(net_arch=net_arch)) if wide else {}
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=0, batch_size=B, n_epochs=E, target_kl=None, ent_coef=ent, **akw)
    sd0 = agent.policy.state_dict()
    assert "log_std" not in sd0 and agent.policy.wide == bool(wide)
    okw = oracle_arch_kwargs(net_arch) if wide else {}
    op = o_nets.TwoCriticPolicy(O, A, discrete=True, **okw)
    op.load_state_dict(sd0)
    obs = rng.randn(T, N, O).astype(np.float32)
    with torch.no_grad():
        a, vr, vc, lp = op.forward(torch.as_tensor(obs.reshape(-1, O)), noise=torch.as_tensor(rng.rand(T * N).astype(np.float32)))
    assert len(np.unique(a.numpy())) == A or A > 8
    # the rollout-side forward / evaluate_actions of the same policy (wide: policy_generic_kernel's Categorical branch) vs the oracle
    unif = rng.rand(T * N).astype(np.float32)
    with torch.no_grad():
        o_a, o_vr, o_vc, o_lp = op.forward(torch.as_tensor(obs.reshape(-1, O)), noise=torch.as_tensor(unif))
        e_vr, e_vc, e_lp, e_ent = op.evaluate_actions(torch.as_tensor(obs.reshape(-1, O)), a)
    h_a, h_vr, h_vc, h_lp = agent.policy.forward(obs.reshape(-1, O), noise=unif)
    assert np.array_equal(h_a.cpu().numpy().ravel().astype(np.int64), o_a.numpy().ravel())
    assert np.allclose(h_lp.cpu().numpy(), o_lp.numpy(), rtol=1e-5, atol=2e-6) and np.allclose(h_vr.cpu().numpy().ravel(), o_vr.numpy().ravel(), rtol=1e-5, atol=2e-6)
    g_vr, g_vc, g_lp, g_ent = agent.policy.evaluate_actions(obs.reshape(-1, O), a.numpy().reshape(-1, 1).astype(np.float32))
    assert np.allclose(g_lp.cpu().numpy(), e_lp.numpy(), rtol=1e-5, atol=2e-6) and np.allclose(g_ent.cpu().numpy(), e_ent.numpy(), rtol=1e-5, atol=2e-6)
    buf = dict(observations=obs, actions=a.numpy().reshape(T, N, 1).astype(np.float32), log_probs=lp.numpy().reshape(T, N),
               reward_values=vr.numpy().reshape(T, N), cost_values=vc.numpy().reshape(T, N),
               reward_advantages=rng.randn(T, N).astype(np.float32) * 2, cost_advantages=rng.rand(T, N).astype(np.float32),
               reward_returns=rng.randn(T, N).astype(np.float32), cost_returns=rng.rand(T, N).astype(np.float32),
               orig_costs=rng.rand(T, N).astype(np.float32))
    rb = agent.rollout_buffer
    for k, v in buf.items():
        getattr(rb, k).copy_(torch.as_tensor(np.asarray(v, np.float32)).reshape(getattr(rb, k).shape))
    rb.full = True
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    nu = agent.dual.nu().item()
    agent.train(perms=perms)
    opt = torch.optim.Adam(op.parameters(), lr=3e-4, eps=1e-5)
    out = o_ppo.ppo_lag_train(op, opt, buf, perms, nu, discrete=True, batch_size=B, n_epochs=E, clip_range=0.2, target_kl=None,
                              ent_coef=ent)
    lg = logger.Logger.CURRENT.name_to_value
    n_steps = agent.policy.adam_step
    for k, v in agent.policy.state_dict().items():
        ref = op.params[k].detach().numpy()
        dev = float(np.abs(v.numpy() - ref).max())
        print(f"ADAM_DEV lgw-oracle {k}: {dev:.3g} = {dev / (3e-4 * n_steps):.3g} x lr x steps")
        assert dev <= ADAM_DEV_BOUND * 3e-4 * n_steps + 2e-7, (k, dev)
    for key in ("train/policy_gradient_loss", "train/reward_value_loss", "train/cost_value_loss", "train/approx_kl",
                "train/clip_fraction", "train/entropy_loss", "train/loss"):
        assert abs(lg[key] - out[key]) < 2e-4 + 2e-3 * abs(out[key]), (key, lg[key], out[key])


def test_sampling_and_evaluation_vs_port():
    """sample_from_agent on LGW-v0 (fixed 200-step episodes, parallel streams) and evaluate_policy on CLGW-v0 (episodes end
    at the first backward move) against the CPU port with the same uniforms."""
    from icrl_amd import utils
    agent, env, cn = _lgw_agent(2, 32, 1)
    sd = agent.policy.state_dict()
    stack = o_loop.make_stack(2, "lgw", 0, norm_obs=False, norm_reward=False, norm_cost=False)
    port = o_loop.PortAgent(stack, n_steps=32, seed=1, discrete=True)
    port.policy.load_state_dict(sd)
    rng = np.random.RandomState(4)
    u = rng.rand(3 * 200).astype(np.float32)
    senv = utils.make_eval_env("LGW-v0", False, normalize_obs=False)
    s1 = o_loop.make_stack(1, "lgw", 0, training=False, norm_obs=False, norm_reward=False, norm_cost=False)
    port.stack = s1
    p_oo, p_o, p_a, p_r, p_l = o_loop.sample_from_agent(port, s1, 3, u)
    for parallel in (False, True):
        oo, o, a, r, l = utils.sample_from_agent(agent, senv, 3, noise=u, parallel=parallel)
        # an action can only differ where the uniform falls within fp32 rounding of the class boundary
        same = a.cpu().numpy().reshape(-1) == p_a.reshape(-1)
        assert list(l) == list(p_l) == [200, 200, 200]
        assert same.mean() > 0.995
        if same.all():
            assert np.array_equal(oo.cpu().numpy(), p_oo) and np.allclose(r, p_r)
    eenv = utils.make_eval_env("CLGW-v0", False, normalize_obs=False)
    e1 = o_loop.make_stack(1, "clgw", 0, training=False, norm_obs=False, norm_reward=False, norm_cost=False)
    port.stack = e1
    u2 = rng.rand(10 * 200).astype(np.float32)
    er, el = utils.evaluate_policy(agent, eenv, 10, deterministic=False, noise=u2, return_episode_rewards=True)
    # port, episode by episode
    k, obs, lens, rews = 0, e1.reset(), [], []
    for ep in range(10):
        done, n, tot = False, 0, 0.0
        while not done:
            act = port.predict(obs, u2[k:k + 1]); k += 1
            obs, r, d, _ = e1.step(act)
            done = bool(d[0]); n += 1; tot += float(r[0])
        lens.append(n); rews.append(tot)
    assert list(el) == lens and np.allclose(er, rews)
    assert min(lens) < 200          # a fresh policy moves backward within a few steps


def test_entry_point_readme_flags_short_run(tmp_path):
    """README.md:25 flags of the reference (LGW-v0 / CLGW-v0, -dno -dnr -dnc, -cl 20, -clr 0.003) at a reduced size."""
    from icrl_amd.icrl import build_parser, icrl
    expert = os.path.join(HERE, "golden/expert_lgw.npz")
    argv = ["icrl", "-er", "20", "-ep", expert, "-tei", "LGW-v0", "-eei", "CLGW-v0", "-tk", "0.01", "-cl", "20", "-clr", "0.003",
            "-ft", "1000", "-ni", "2", "-bi", "5", "-dno", "-dnr", "-dnc", "-nt", "2", "--n_steps", "250", "-s", "0",
            "--expert_agent_path", expert, "--save_dir", str(tmp_path), "-v", "0"]
    cfg = vars(build_parser().parse_args(argv))
    cfg.update(rank=0, world_size=1)
    metrics, agent, cn, env = icrl(types.SimpleNamespace(**cfg), log=None)
    m = metrics[-1]
    assert agent.policy.discrete and cn.is_discrete and cn.input_dims == 3
    for k in ("true/reward", "true/cost", "true/forward_kl", "true/reverse_kl", "forward/nu", "forward/entropy_loss",
              "backward/cn_loss", "backward/expert_loss"):
        assert k in m and np.isfinite(m[k]), k
    assert 0.0 <= m["true/cost"] <= 1.0
