"""GPU parity of the rollout-time kernels (policy forward, env step, cost net, normaliser, fused collect) vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import loop as o_loop, nets as o_nets, stats as o_stats
from oracle.synth_env import SynthVecEnv

pytestmark = pytest.mark.gpu


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


@pytest.mark.parametrize("kind,wall,broken", [("hc", False, False), ("ant", False, False), ("hc", True, False), ("ant", False, True)])
def test_synth_env_bit_exact(kind, wall, broken):
    from icrl_amd.vec_env import HipSynthVecEnv
    N, steps = 7, 1100 if kind == "hc" else 600
    cpu = SynthVecEnv(N, kind, seed=3, wall_terminate=wall, broken=broken)
    gpu = HipSynthVecEnv(N, kind, seed=3, wall_terminate=wall, broken=broken)
    assert np.array_equal(cpu.reset(), gpu.reset().cpu().numpy())
    rng = np.random.RandomState(0)
    for t in range(steps):
        a = rng.uniform(-1, 1, (N, cpu.act_dim)).astype(np.float32)
        if wall and t % 50 < 25:
            a = -np.sign(cpu.B[0])[None].repeat(N, 0).astype(np.float32)      # drive obs[0] towards the wall
        o, r, d = cpu.step(a)
        go, gr, gd, _ = gpu.step(a)
        assert np.array_equal(o, go.cpu().numpy()), t
        assert np.array_equal(r, gr.cpu().numpy()), t
        assert np.array_equal(d, gd.cpu().numpy().astype(bool)), t
    if wall:
        assert cpu.step_count.max() > 0


@pytest.mark.parametrize("case", ["hc", "ant", "lgw", "antbroken", "hc_norm"])
def test_cost_function_golden(golden, case):
    from icrl_amd.constraint_net import ConstraintNet
    g = _sub(golden("g2_cost_function"), case + "/")
    disc = case == "lgw"
    od, ad = g["obs"].shape[1], (2 if disc else g["acs"].shape[1])
    kw = dict(clip_obs=20, action_low=None if disc else -np.ones(ad, np.float32), action_high=None if disc else np.ones(ad, np.float32))
    if case == "antbroken":
        kw = dict(clip_obs=None, action_low=None, action_high=None)
    if case == "hc_norm":        # --cn_normalize: (obs - mean) / sqrt(var + 1e-5) before the clip (constraint_net.py:275-283)
        kw.update(initial_obs_mean=g["mean"], initial_obs_var=g["var"])
    cn = ConstraintNet(od, ad, list(g["hidden"]), None, lambda x: 0.05, None, None, disc, **kw)
    assert cn.select_dim == list(g["select_dim"])
    cn.load_state_dict({k[2:]: g[k] for k in g if k.startswith("w/")})
    got = cn.cost_function(g["obs"], g["acs"])
    # fp32 MLP: summation order / expf differ from torch-CPU by a few ulp
    assert np.allclose(got, g["cost"], rtol=2e-5, atol=2e-6)


def test_vecnormalize_stream_golden(golden):
    """float64 statistics reproduce numpy's reduction order: bit-exact against the reference stream."""
    from icrl_amd.vec_env import VecEnv, VecCostWrapper, VecNormalizeWithCost, BatchedInfos
    from icrl_amd import spaces
    g = golden("g3_vecnormalize")
    S, N, D = g["obs"].shape
    dev = torch.device("cuda")

    class Replay(VecEnv):
        def __init__(self):
            super().__init__(N, spaces.Box(-np.inf, np.inf, (D,), np.float64), spaces.Box(-1, 1, (2,), np.float32))
            self.t, self.device = 0, dev
        def reset(self): return torch.as_tensor(g["reset_obs"], device=dev)
        def step_async(self, a): pass
        def step_wait(self):
            t = self.t; self.t += 1
            return (torch.as_tensor(g["obs"][t], device=dev), torch.as_tensor(g["rew"][t], device=dev),
                    torch.as_tensor(g["done"][t].astype(np.uint8), device=dev), BatchedInfos(N))
    venv = VecCostWrapper(Replay())
    step = {"t": 0}
    venv.set_cost_function(lambda o, a: g["costs"][step["t"]])
    env = VecNormalizeWithCost(venv, reward_gamma=float(g["gammas"][0]), cost_gamma=float(g["gammas"][1]))
    assert np.array_equal(env.reset().cpu().numpy(), g["reset_obs_n"])
    for t in range(S):
        step["t"] = t
        o, r, d, infos = env.step(np.zeros((N, 2), np.float32))
        assert np.array_equal(o.cpu().numpy(), g["obs_n"][t]), t
        assert np.array_equal(r.cpu().numpy(), g["rew_n"][t]), t
        assert np.array_equal(infos.batch["cost"].cpu().numpy(), g["cost_n"][t]), t
        assert np.array_equal(env.obs_rms.var, g["obs_var"][t]) and env.ret_rms.var == g["ret_var"][t]
        assert env.cost_rms.var == g["cost_var"][t] and env.cost_rms.count == g["cost_count"][t]
        assert np.array_equal(env.get_original_cost().cpu().numpy(), g["costs"][t])


@pytest.mark.parametrize("N", [1, 8, 127, 128, 129, 300, 1000, 2048, 3000, 4096])
def test_vecnormalize_numpy_reduction_order(N):
    from icrl_amd.vec_env import HipSynthVecEnv, VecNormalizeWithCost, VecCostWrapper
    rng = np.random.RandomState(N)
    st = o_stats.NormState(N, 18)
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "hc")))
    nm_obs = torch.zeros(N, 18, dtype=torch.float64, device="cuda")
    o_stats.norm_reset(st, np.zeros((N, 18)))
    env.reset()
    for t in range(5):
        obs, rew = rng.randn(N, 18) * 7 + 1, rng.randn(N) * 3
        cost, done = rng.rand(N).astype(np.float32), rng.rand(N) < 0.2
        oo, ro, co = o_stats.norm_step(st, obs, rew, cost, done)
        dev = lambda x, dt: torch.as_tensor(x, device="cuda").to(dt).contiguous()
        env._norm_call(dev(obs, torch.float64), dev(rew, torch.float64), dev(cost, torch.float32), dev(done, torch.uint8))
        assert np.array_equal(env._obs_out.cpu().numpy(), oo) and np.array_equal(env._rew_out.cpu().numpy(), ro)
        assert np.array_equal(env._cost_out.cpu().numpy(), co)
        assert np.array_equal(env.ret.cpu().numpy(), st.ret) and env.ret_rms.var == st.ret_rms.var


def test_policy_forward_vs_oracle(golden):
    from icrl_amd.policies import ActorTwoCriticsPolicy
    from icrl_amd import spaces
    g = golden("g9_learn_iteration")
    pol = ActorTwoCriticsPolicy(spaces.Box(-np.inf, np.inf, (18,), np.float64), spaces.Box(-1, 1, (6,), np.float32))
    pol.load_state_dict(_sub(g, "w0/"))
    assert all(torch.equal(v, torch.as_tensor(g["w0/" + k])) for k, v in pol.state_dict().items())
    op = o_nets.TwoCriticPolicy(18, 6); op.load_state_dict(_sub(g, "w0/"))
    rng = np.random.RandomState(1)
    obs, noise = rng.randn(33, 18) * 2, rng.randn(33, 6).astype(np.float32)
    a, vr, vc, lp = pol.forward(obs, noise=noise)
    with torch.no_grad():
        oa, ovr, ovc, olp = op.forward(torch.as_tensor(obs), torch.as_tensor(noise))
    for got, ref in ((a, oa), (vr, ovr), (vc, ovc), (lp, olp)):
        assert np.allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6)
    assert np.allclose(pol.last_clipped.cpu().numpy(), np.clip(oa.numpy(), -1, 1), atol=2e-6)
    a_det = pol.forward(obs, deterministic=True)[0]
    with torch.no_grad():
        assert np.allclose(a_det.cpu().numpy(), op.forward(torch.as_tensor(obs), deterministic=True)[0].numpy(), atol=2e-6)


def _build_stack(N, kind, cn_sd, seed=0):
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    from icrl_amd.constraint_net import ConstraintNet
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, seed)), reward_gamma=0.99, cost_gamma=0.99)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    cn = ConstraintNet(od, ad, [20], None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20,
                       action_low=-np.ones(ad, np.float32), action_high=np.ones(ad, np.float32), per_step_importance_sampling=True)
    cn.load_state_dict(cn_sd)
    env.set_cost_function(cn.cost_function)
    return env, cn


def test_fused_rollout_vs_reference_golden(golden):
    """icrl_rollout_collect vs the REFERENCE's own first rollout (tests/golden/g9, N=4, T=32), teacher-forced noise.
    Tolerance: fp32 policy / cost MLPs differ from torch-CPU in summation order (1e-5 relative on activations)."""
    from icrl_amd.ppo_lag import PPOLagrangian
    g = golden("g9_learn_iteration")
    noise = g["noise"]
    _, T, N, A = noise.shape
    env, cn = _build_stack(N, "hc", _sub(g, "cn/"))
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, seed=0)
    agent.policy.load_state_dict(_sub(g, "w0/"))
    agent._setup_learn(2 * N * T)
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=torch.as_tensor(noise[0], device="cuda").contiguous())
    # the golden buffer holds the SECOND rollout (the reference re-uses the buffer); rebuild the first from the oracle port
    stack = o_loop.make_stack(N, "hc", 0)
    lo = -np.ones(6, np.float32)
    ocn = o_nets.CostNet(18, 6, [20], False, None, None, 20, lo, -lo); ocn.load_state_dict(_sub(g, "cn/"))
    stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, seed=0)
    port.policy.load_state_dict(_sub(g, "w0/"))
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    b = port.collect_rollouts(noise[0])
    rb = agent.rollout_buffer
    for k in ("observations", "orig_observations", "new_observations", "new_orig_observations", "actions", "rewards", "costs",
              "orig_costs", "dones", "log_probs", "reward_values", "cost_values", "reward_advantages", "cost_advantages",
              "reward_returns", "cost_returns"):
        got, ref = getattr(rb, k).cpu().numpy().reshape(T, N, -1), getattr(b, k).reshape(T, N, -1)
        assert np.allclose(got, ref, rtol=2e-4, atol=2e-5), (k, np.abs(got - ref).max())
    assert np.allclose(env.obs_rms.mean, stack.norm.obs_rms.mean, rtol=1e-6, atol=1e-7)
    assert np.allclose(env.obs_rms.var, stack.norm.obs_rms.var, rtol=1e-6, atol=1e-9)
    assert abs(env.cost_rms.var - stack.norm.cost_rms.var) < 1e-6 * max(1.0, stack.norm.cost_rms.var)
    assert agent.num_timesteps == N * T


# ant 96 / hc 256 / antbroken 512: per-step launches (generic normaliser kernel); the others: persistent kernel.
# hc 256 = the per-GPU shard of BASELINE configs[3] (2048 envs / 8), antbroken 512 = that of configs[4] (4096 envs / 8) with the
# reference's committed AntBroken constraint net loaded through the (quirky) ConstraintNet.load and action[4:] zeroed.
# kernel "multi": rollout_multi_kernel (MFMA tiles, several envs per workgroup: every seed-batch rollout and any run with more than 1024
# envs) forced at shapes the other kernels normally serve, so that it is compared with the ORACLE directly and not only with the
# per-step launches (on_policy_algorithm.py:340-421, policies.py:716-731).
@pytest.mark.parametrize("kind,N,T,kernel", [("hc", 64, 300, "auto"), ("ant", 96, 40, "auto"), ("ant", 32, 60, "auto"), ("hc", 256, 40, "auto"),
                                             ("antbroken", 512, 12, "auto"), ("hc", 64, 300, "multi"), ("antbroken", 512, 12, "multi"),
                                             ("ant", 96, 40, "multi"),
                                             # a policy with layers above 64 (-pl 128 96 -rvl 80 128 -cvl 128 128, icrl/utils.py:636-655): the
                                             # per-step launches of icrl_rollout_collect with the generic-shape forward kernel
                                             ("hc", 16, 120, "wide-policy"), ("ant", 8, 30, "wide-policy")])
def test_fused_rollout_vs_port(kind, N, T, kernel):
    """same comparison at HC / Ant shapes with freshly initialised nets; also crosses episode ends (hc T=300 < 1000: none,
    so the env is pre-stepped to t_ep = 900 first)."""
    import os
    from icrl_amd.ppo_lag import PPOLagrangian
    torch.manual_seed(5)
    broken = kind == "antbroken"
    ekind = "ant" if broken else kind
    od, ad = (18, 6) if ekind == "hc" else (113, 8)
    hid = [20] if ekind == "hc" else [40, 40]
    lo = -np.ones(ad, np.float32)
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    from icrl_amd.constraint_net import ConstraintNet
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, ekind, 7, broken=broken)))
    if broken:
        cn = ConstraintNet.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden/cn_antbroken.npz"))
        assert cn.clip_obs is None and cn.action_low is None and list(cn.hidden_sizes) == [40, 40]
        ocn = o_nets.CostNet(od, ad, hid, False, None, None, None, None, None)      # what the off-by-one load() leaves
        ocn.load_state_dict(cn.state_dict())
    else:
        ocn = o_nets.CostNet(od, ad, hid, False, None, None, 20, lo, -lo)
        cn = ConstraintNet(od, ad, hid, None, lambda x: 0.05, None, None, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo)
        cn.load_state_dict(ocn.state_dict())
    env.set_cost_function(cn.cost_function)
    wide = kernel == "wide-policy"
    akw = dict(policy_kwargs=dict(net_arch=[dict(pi=[128, 96], vf=[80, 128], cvf=[128, 128])])) if wide else {}
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=7, **akw)   # learn() re-seeds the env with the agent seed
    if wide:
        assert agent.policy.wide and agent._fused_chain() is not None
    else:
        agent.rollout_kernel = kernel
    stack = o_loop.make_stack(N, ekind, 7, broken=broken); stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, seed=7, **(dict(hidden=dict(policy_net=(128, 96), value_net=(80, 128), cost_value_net=(128, 128))) if wide else {}))
    port.policy.load_state_dict(agent.policy.state_dict())        # same seed gives the same init; make it explicit anyway
    near_end = {"hc": 1000 - T // 3, "antbroken": 500 - T // 3}.get(kind)      # cross an episode end inside the rollout
    rng = np.random.RandomState(2)
    noise = rng.randn(T, N, ad).astype(np.float32)
    agent._setup_learn(N * T)
    if near_end is not None:
        env.unwrapped.t_ep.fill_(near_end)
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    if near_end is not None:
        stack.env.t_ep[:] = near_end
    agent.collect_rollouts(env, None, agent.rollout_buffer, T, "cost", noise=torch.as_tensor(noise, device="cuda"))
    agent.check_rollout_status()
    b = port.collect_rollouts(noise)
    rb = agent.rollout_buffer
    if near_end is not None:
        assert b.dones.sum() == N            # every env crossed an episode end
    if broken:                               # ant.py:105-108 — the env ignores action[4:]; the buffer keeps what the policy drew
        assert np.abs(rb.actions.cpu().numpy()[..., 4:]).max() > 0
    for k in ("observations", "orig_observations", "new_observations", "actions", "rewards", "costs", "orig_costs", "dones",
              "log_probs", "reward_values", "cost_values", "reward_advantages", "cost_advantages", "reward_returns", "cost_returns"):
        got, ref = getattr(rb, k).cpu().numpy().reshape(T, N, -1), getattr(b, k).reshape(T, N, -1)
        assert np.allclose(got, ref, rtol=5e-4, atol=5e-5), (k, np.abs(got - ref).max())
    # float64 running moments over N x T samples (the samples themselves carry the fp32 action differences)
    assert np.allclose(env.obs_rms.mean, stack.norm.obs_rms.mean, rtol=1e-5, atol=1e-6)
    assert np.allclose(env.obs_rms.var, stack.norm.obs_rms.var, rtol=1e-5, atol=1e-8)
    assert abs(env.ret_rms.var - stack.norm.ret_rms.var) <= 1e-5 * max(1.0, stack.norm.ret_rms.var)
    assert env.obs_rms.count == stack.norm.obs_rms.count


def _pair_of_agents(N, T, seed, kind="hc", broken=False, cn_kwargs=None, hid=None, agent_kwargs=None):
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    from icrl_amd.constraint_net import ConstraintNet
    out = []
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    hid = hid or ([20] if kind == "hc" else [40, 40])
    lo = -np.ones(ad, np.float32)
    for _ in range(2):
        torch.manual_seed(seed)
        env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, kind, seed, broken=broken)))
        kw = dict(clip_obs=20, action_low=lo, action_high=-lo) if cn_kwargs is None else cn_kwargs
        cn = ConstraintNet(od, ad, hid, None, lambda x: 0.05, None, None, False, 0.5, **kw)
        env.set_cost_function(cn.cost_function)
        out.append((PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=seed, **(agent_kwargs or {})), env, cn))
    out[1][2].load_state_dict(out[0][2].state_dict())
    out[1][0].policy.load_state_dict(out[0][0].policy.state_dict())
    return out


_BUF_KEYS = ("observations", "orig_observations", "new_observations", "new_orig_observations", "actions", "rewards", "costs",
             "orig_costs", "dones", "log_probs", "reward_values", "cost_values", "reward_advantages", "cost_advantages",
             "reward_returns", "cost_returns")


def test_stepped_rollout_equals_fused_rollout():
    """the per-step loop over the fine-grained entry points (the reference's own structure, on_policy_algorithm.py:340-421)
    and the fused launch sequence are two drivers of the same kernels: identical buffers and normaliser statistics."""
    N, T = 16, 48
    (a_f, e_f, _), (a_s, e_s, _) = _pair_of_agents(N, T, 11)
    noise = torch.as_tensor(np.random.RandomState(3).randn(2, T, N, 6).astype(np.float32), device="cuda")
    a_f._setup_learn(2 * N * T); a_s._setup_learn(2 * N * T)
    for it in range(2):
        a_f.collect_rollouts(e_f, None, a_f.rollout_buffer, T, "cost", noise=noise[it])
        a_s._collect_rollouts_stepped(e_s, None, a_s.rollout_buffer, T, "cost", noise=noise[it])
        for k in _BUF_KEYS:
            got, ref = getattr(a_s.rollout_buffer, k).cpu().numpy(), getattr(a_f.rollout_buffer, k).cpu().numpy()
            assert np.allclose(got, ref, rtol=2e-5, atol=2e-6), (it, k, np.abs(got - ref).max())
    assert np.allclose(e_s.obs_rms.mean, e_f.obs_rms.mean, rtol=0, atol=1e-12) and abs(e_s.cost_rms.var - e_f.cost_rms.var) < 1e-12
    assert a_s.num_timesteps == a_f.num_timesteps == 2 * N * T


@pytest.mark.parametrize("kind,shape,N", [("hc", "wide", 12), ("hc", "trunk", 12), ("ant", "deep", 12), ("hc", "wide-cn", 12), ("ant", "deep-cn", 12), ("hc", "both", 12),
                                          ("hc", "trunk", 64), ("ant", "deep", 33)])
def test_generic_shape_rollout_equals_python_loop(kind, shape, N):
    """policies / constraint nets of the generic-shape path (layers above 64 units, shared trunk, other depths): icrl_rollout_collect runs the
    whole rollout as ONE persistent launch (rollout_generic_kernel: the one-workgroup-per-env loop around the table-driven forward with four
    units per lane) — or, for a constraint net beyond the register image, the reference's per-step loop as four launches per step —
    (csrc/rollout.hip) and must leave exactly what the Python loop over the fine-grained entry points leaves: every unit's value is the same
    chain of fused multiply-adds in the same order."""
    from helpers.arches import ARCHES
    T = 40
    net_arch = {"wide": [dict(pi=[128, 96], vf=[80, 128], cvf=[128, 128])], "both": [dict(pi=[128, 96], vf=[80, 128], cvf=[128, 128])]}.get(shape, ARCHES.get(shape))
    hid = {"wide-cn": [128, 100], "deep-cn": [48, 32, 24], "both": [64, 64, 64]}.get(shape)
    akw = dict(policy_kwargs=dict(net_arch=net_arch)) if net_arch else None
    (a_f, e_f, c_f), (a_s, e_s, _) = _pair_of_agents(N, T, 11, kind=kind, hid=hid, agent_kwargs=akw)
    assert a_f.policy.wide == (net_arch is not None) and c_f.wide == (hid is not None) and a_f._fused_chain() is not None
    ad = 6 if kind == "hc" else 8
    noise = torch.as_tensor(np.random.RandomState(3).randn(2, T, N, ad).astype(np.float32), device="cuda")
    a_f._setup_learn(2 * N * T); a_s._setup_learn(2 * N * T)
    for it in range(2):
        a_f.collect_rollouts(e_f, None, a_f.rollout_buffer, T, "cost", noise=noise[it])
        a_s._collect_rollouts_stepped(e_s, None, a_s.rollout_buffer, T, "cost", noise=noise[it])
        for k in _BUF_KEYS:
            got, ref = getattr(a_s.rollout_buffer, k).cpu().numpy(), getattr(a_f.rollout_buffer, k).cpu().numpy()
            assert np.array_equal(got, ref), (it, k, np.abs(got - ref).max())
    assert np.array_equal(e_s.obs_rms.mean, e_f.obs_rms.mean) and e_s.cost_rms.var == e_f.cost_rms.var and e_s.ret_rms.var == e_f.ret_rms.var
    assert a_s.num_timesteps == a_f.num_timesteps == 2 * N * T


def test_callable_cost_function_warmup_semantics():
    """learn(cost_function=null_cost) (ref: icrl/icrl.py:187-193, on_policy_algorithm.py:392-394): the env chain still steps
    with the cost net (cost_rms keeps updating), the buffer's costs come from the callable on (obs AFTER the step, clipped
    action), un-normalised; a second callable checks that pairing."""
    from icrl_amd.true_constraint_net import null_cost
    N, T = 8, 40
    (a_f, e_f, _), (a_s, e_s, _) = _pair_of_agents(N, T, 5)
    noise = torch.as_tensor(np.random.RandomState(4).randn(T, N, 6).astype(np.float32), device="cuda")
    a_f._setup_learn(N * T); a_s._setup_learn(N * T)
    a_f.collect_rollouts(e_f, None, a_f.rollout_buffer, T, "cost", noise=noise)
    seen = []

    def probe(obs, acs):
        seen.append((obs.copy(), acs.copy()))
        return (obs[..., 0] > 0).astype(np.float32) + np.abs(acs).max(-1)

    a_s.collect_rollouts(e_s, None, a_s.rollout_buffer, T, probe, noise=noise)
    rf, rs = a_f.rollout_buffer, a_s.rollout_buffer
    for k in ("observations", "new_orig_observations", "actions", "rewards", "log_probs", "reward_values", "cost_values",
              "reward_advantages"):
        assert np.allclose(getattr(rs, k).cpu().numpy(), getattr(rf, k).cpu().numpy(), rtol=2e-5, atol=2e-6), k
    assert abs(e_s.cost_rms.var - e_f.cost_rms.var) < 1e-12          # the wrapper chain saw the cost net's costs
    new_orig = rs.new_orig_observations.cpu().numpy()
    clipped = np.clip(rs.actions.cpu().numpy(), -1, 1)
    want = (new_orig[..., 0] > 0).astype(np.float32) + np.abs(clipped).max(-1)
    assert np.allclose(rs.costs.cpu().numpy(), want, atol=1e-6) and np.array_equal(rs.costs.cpu().numpy(), rs.orig_costs.cpu().numpy())
    assert len(seen) == T and seen[0][0].shape == (N, 18) and seen[0][1].shape == (N, 6)
    assert np.abs(seen[3][1]).max() <= 1.0
    # cost GAE of the callable's costs
    from oracle.gae import dual_gae
    g = dual_gae(rs.rewards.cpu().numpy(), rs.costs.cpu().numpy(), rs.reward_values.cpu().numpy(), rs.cost_values.cpu().numpy(),
                 rs.dones.cpu().numpy(), rs.reward_values.cpu().numpy()[-1],
                 rs.cost_values.cpu().numpy()[-1], a_s._last_dones.cpu().numpy().astype(bool), 0.99, 0.95, 0.99, 0.95)
    assert np.allclose(rs.cost_advantages.cpu().numpy(), g["cost_advantages"], rtol=1e-5, atol=1e-5)
    # null_cost through learn(): zero costs, a full update runs
    a_s.learn(N * T, cost_function=null_cost)
    assert float(a_s.rollout_buffer.costs.abs().max().item()) == 0.0 and float(a_s.rollout_buffer.orig_costs.abs().max().item()) == 0.0
    assert a_s.policy.adam_step > 0


def test_analytic_env_cost_through_cost_wrapper():
    """cpg without --cn_path (ref: icrl/cpg.py:52-53,109): the env cost is the analytic wall_behind(-3) on (previous raw obs,
    action); it flows through VecCostWrapper's callable branch, cost normalisation and the stepped rollout."""
    from icrl_amd.ppo_lag import PPOLagrangian
    from icrl_amd.true_constraint_net import get_true_cost_function
    from icrl_amd.vec_env import HipSynthVecEnv, VecCostWrapper, VecNormalizeWithCost
    N, T = 4, 30
    env = VecNormalizeWithCost(VecCostWrapper(HipSynthVecEnv(N, "hc", 2)), norm_cost=False)
    env.set_cost_function(lambda o, a: (o[..., 0] <= 0.0))       # threshold 0 so that both outcomes occur on this env
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, seed=2)
    agent.learn(N * T, cost_function="cost")
    rb = agent.rollout_buffer
    want = (rb.orig_observations.cpu().numpy()[..., 0] <= 0.0).astype(np.float32)
    assert np.array_equal(rb.orig_costs.cpu().numpy(), want) and 0.0 < want.mean() < 1.0
    assert get_true_cost_function("HCWithPosTest-v0")(np.array([[-3.5, 0.0]]), None)[0]


# N <= 128: rollout_persistent_kernel (statistics replicated per workgroup); beyond: rollout_wide_kernel (statistics partitioned by
# observation column, two hops per step; 512 envs = two envs per workgroup).  hc 256 / ant 256 / antbroken 512 are the per-GPU
# shards of BASELINE configs[3], [2] and [4].
@pytest.mark.parametrize("kind,N,T", [("hc", 64, 300), ("hc", 7, 33), ("hc", 128, 20), ("hc", 256, 40), ("hc", 130, 24), ("ant", 256, 12),
                                      ("antbroken", 512, 10), ("hc", 1000, 6)])
def test_persistent_rollout_equals_per_step_launches(kind, N, T, kernel="auto", cn_kwargs=None):
    """the one-launch rollout (device-wide exchange per step inside the kernel) against the launch pair per
    step: every buffer plane, the normaliser state and the agent's carry-over state are bit-identical, across two
    consecutive rollouts and across episode ends."""
    ekind = "ant" if kind == "antbroken" else kind
    ad = 6 if ekind == "hc" else 8
    limit = 1000 if ekind == "hc" else 500
    (a_p, e_p, _), (a_s, e_s, _) = _pair_of_agents(N, T, 13, ekind, broken=kind == "antbroken", cn_kwargs=cn_kwargs)
    a_s.rollout_kernel = "steps"
    a_p.rollout_kernel = kernel
    noise = torch.as_tensor(np.random.RandomState(8).randn(2, T, N, ad).astype(np.float32), device="cuda")
    a_p._setup_learn(2 * N * T); a_s._setup_learn(2 * N * T)
    for env in (e_p, e_s):
        env.unwrapped.t_ep.fill_(limit - T // 2)           # every env crosses its time limit inside the first rollout
    for it in range(2):
        a_p.collect_rollouts(e_p, None, a_p.rollout_buffer, T, "cost", noise=noise[it])
        a_s.collect_rollouts(e_s, None, a_s.rollout_buffer, T, "cost", noise=noise[it])
        a_p.check_rollout_status()
        for k in _BUF_KEYS:
            got, ref = getattr(a_p.rollout_buffer, k).cpu().numpy(), getattr(a_s.rollout_buffer, k).cpu().numpy()
            assert np.array_equal(got, ref), (it, k, np.abs(got - ref).max())
        assert a_p.rollout_buffer.dones.sum().item() == (N if it == 0 else a_s.rollout_buffer.dones.sum().item())
    for name in ("obs_rms", "ret_rms", "cost_rms"):
        rp, rs = getattr(e_p, name), getattr(e_s, name)
        assert np.array_equal(np.asarray(rp.mean), np.asarray(rs.mean)) and np.array_equal(np.asarray(rp.var), np.asarray(rs.var))
        assert rp.count == rs.count
    assert torch.equal(e_p.ret, e_s.ret) and torch.equal(e_p.cost_ret, e_s.cost_ret)
    assert torch.equal(a_p._last_obs, a_s._last_obs) and torch.equal(a_p._ag["last_dones"], a_s._ag["last_dones"])
    assert torch.equal(e_p.unwrapped.s, e_s.unwrapped.s) and torch.equal(e_p.unwrapped.t_ep, e_s.unwrapped.t_ep)
    for k in ("raw_rew", "raw_cost", "dones", "last_v_r", "last_v_c", "act_clipped"):
        assert torch.equal(a_p._ag[k], a_s._ag[k]), k


# rollout_multi_kernel: E = 4 or 8 envs per workgroup evaluated interleaved, statistics owners on all four waves.  hc 64 = the
# shape of a seed batch's runs (8 workgroups x 8 envs), hc 20 / 37 = ragged last workgroups (E = 4), hc 300 / ant 256 / antbroken 512 =
# the many-environment shapes (ant: 29+ workgroups needed for the 115 statistics)
@pytest.mark.parametrize("kind,N,T", [("hc", 64, 300), ("hc", 20, 33), ("hc", 37, 21), ("hc", 128, 20), ("hc", 300, 12), ("ant", 256, 12),
                                      ("antbroken", 512, 10), ("hc", 1000, 6),
                                      # beyond 1024 envs per GPU (BASELINE configs[3] / configs[4] whole on ONE GPU): 8 / 16 envs per
                                      # workgroup, 256 workgroups; the per-step normaliser handles up to 4096 envs as well
                                      ("hc", 2048, 5), ("antbroken", 4096, 3)])
def test_multi_env_rollout_equals_per_step_launches(kind, N, T):
    test_persistent_rollout_equals_per_step_launches(kind, N, T, kernel="multi")


@pytest.mark.parametrize("kernel", ["multi", "auto"])
@pytest.mark.parametrize("kind,N,T", [("hc", 64, 40), ("ant", 128, 10)])
def test_rollout_kernels_with_a_normalising_unclipped_constraint_net(kind, N, T, kernel):
    """the other branches of ConstraintNet.prepare_data inside the fused rollouts: --cn_normalize statistics (mean / var per
    observation column; the multi-env kernel keeps mean and sqrt(var + eps) of its input columns in registers), no observation
    clipping, no action bounds (what ConstraintNet.load builds, constraint_net.py:394-399)."""
    od = 18 if kind == "hc" else 113
    rng = np.random.RandomState(4)
    kw = dict(clip_obs=None, initial_obs_mean=rng.randn(od) * 0.1, initial_obs_var=rng.rand(od) + 0.5)
    test_persistent_rollout_equals_per_step_launches(kind, N, T, kernel=kernel, cn_kwargs=kw)
