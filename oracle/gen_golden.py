"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference):

    PYTHONPATH=oracle/ref_shim:/root/reference:/root/reference/custom_envs python -W ignore -m oracle.gen_golden

The reference is imported unmodified under the third-party stand-ins of ``oracle/ref_shim`` (gym /
wandb / mpl_scatter_density are not installed here).  Every golden file stores
inputs + the reference's outputs (+ torch/numpy versions); at generation time the oracle restatement
is checked against the reference on the same inputs and the max deviation is printed.
Nothing of the reference's source is written anywhere — only arrays.
"""
import io
import os
import pickle
import sys
import zipfile

import numpy as np
import torch as th

_orig_load = th.load
th.load = lambda f, **k: _orig_load(f, **{**k, "weights_only": False})   # reference checkpoints hold numpy arrays

import gym  # noqa: E402  (the shim)
from stable_baselines3 import PPOLagrangian  # noqa: E402
from stable_baselines3.common import logger as ref_logger  # noqa: E402
from stable_baselines3.common.buffers import RolloutBufferWithCost  # noqa: E402
from stable_baselines3.common.dual_variable import DualVariable  # noqa: E402
from stable_baselines3.common.vec_env import VecCostWrapper, VecNormalizeWithCost  # noqa: E402
from stable_baselines3.common.vec_env.base_vec_env import VecEnv  # noqa: E402
from icrl.constraint_net import ConstraintNet  # noqa: E402
import icrl.utils as ref_utils  # noqa: E402

from oracle import cn as o_cn, gae as o_gae, loop as o_loop, nets as o_nets, ppo as o_ppo, stats as o_stats  # noqa: E402
from oracle.synth_env import SynthVecEnv  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
META = dict(torch=th.__version__, numpy=np.__version__)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: np.asarray(v) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(repr(META)), **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KB)")


def maxdiff(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


class RefSynthVecEnv(VecEnv):
    """The synthetic env behind the reference's own VecEnv ABC (ref: vec_env/base_vec_env.py:48-224).
    Returns float64 obs / rewards like SubprocVecEnv does (np.stack of the workers' results)."""

    def __init__(self, n_envs, kind="hc", seed=0, **kw):
        self.e = SynthVecEnv(n_envs, kind, seed, **kw)
        super().__init__(n_envs, gym.spaces.Box(-np.inf, np.inf, (self.e.obs_dim,), np.float64),
                         gym.spaces.Box(-1, 1, (self.e.act_dim,), np.float32))

    def reset(self):
        return self.e.reset()

    def step_async(self, actions):
        self._a = actions

    def step_wait(self):
        o, r, d = self.e.step(self._a)
        return o, r, d, [{} for _ in range(self.num_envs)]

    def close(self): pass
    def seed(self, seed=None): return [seed]
    def get_attr(self, n, indices=None): return [getattr(self.e, n)]
    def set_attr(self, *a, **k): pass
    def env_method(self, *a, **k): return []


# ------------------------------------------------------------------------------------------------
def g1_gae():
    print("G1 dual GAE")
    rng = np.random.RandomState(1)
    cases = {}
    for name, (T, N, gr, lr_, gc, lc) in dict(t8n3=(8, 3, 0.99, 0.95, 0.99, 0.95), t2000n1=(2000, 1, 0.99, 0.95, 0.99, 0.95),
                                               t256n16=(256, 16, 0.99, 0.9, 0.99, 0.9), t1n4=(1, 4, 0.99, 0.95, 0.97, 0.8),
                                               t64n5=(64, 5, 0.99, 0.95, 0.99, 0.9)).items():
        space_o, space_a = gym.spaces.Box(-1, 1, (3,), np.float32), gym.spaces.Box(-1, 1, (2,), np.float32)
        b = RolloutBufferWithCost(T, space_o, space_a, "cpu", gr, lr_, gc, lc, n_envs=N)
        b.rewards[:] = rng.randn(T, N); b.costs[:] = rng.rand(T, N) * 3
        b.reward_values[:] = rng.randn(T, N); b.cost_values[:] = rng.randn(T, N)
        b.dones[:] = (rng.rand(T, N) < 0.05)
        lv_r, lv_c = th.tensor(rng.randn(N, 1), dtype=th.float32), th.tensor(rng.randn(N, 1), dtype=th.float32)
        ld = rng.rand(N) < 0.3
        b.compute_returns_and_advantage(lv_r, lv_c, ld)
        o = o_gae.dual_gae(b.rewards, b.costs, b.reward_values, b.cost_values, b.dones, lv_r.numpy(), lv_c.numpy(), ld,
                           gr, lr_, gc, lc)
        for k in o:
            assert np.array_equal(o[k], getattr(b, k)), (name, k, maxdiff(o[k], getattr(b, k)))
            assert getattr(b, k).dtype == np.float32
        for k in ("rewards", "costs", "reward_values", "cost_values", "dones", "reward_returns", "reward_advantages",
                  "cost_returns", "cost_advantages"):
            cases[f"{name}/{k}"] = getattr(b, k).copy()
        cases[f"{name}/last_v_r"], cases[f"{name}/last_v_c"], cases[f"{name}/last_dones"] = lv_r.numpy().ravel(), lv_c.numpy().ravel(), ld
        cases[f"{name}/params"] = np.array([gr, lr_, gc, lc])
    print("  oracle == reference bit-for-bit on all cases")
    save("g1_gae", **cases)


def _cn_kwargs(obs_dim, acs_dim, hidden, expert_obs, expert_acs, **kw):
    d = dict(obs_dim=obs_dim, acs_dim=acs_dim, hidden_sizes=hidden, batch_size=None, lr_schedule=lambda x: 0.05,
             expert_obs=expert_obs, expert_acs=expert_acs, is_discrete=False, regularizer_coeff=0.5,
             clip_obs=20, action_low=-np.ones(acs_dim, np.float32), action_high=np.ones(acs_dim, np.float32),
             target_kl_old_new=10, target_kl_new_old=2.5)
    d.update(kw)
    return d


def _sd_np(sd, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in sd.items()}


def g2_cost_function():
    print("G2 ConstraintNet.cost_function")
    rng = np.random.RandomState(2)
    out = {}
    for name, (od, ad, hid, disc) in dict(hc=(18, 6, [20], False), ant=(113, 8, [40, 40], False), lgw=(1, 2, [20], True)).items():
        th.manual_seed(3)
        kw = _cn_kwargs(od, ad, hid, None, None, is_discrete=disc)
        if disc:
            kw.update(action_low=None, action_high=None)
        ref = ConstraintNet(**kw)
        obs = rng.randn(37, od) * 15                      # beyond clip_obs = 20 in places
        acs = rng.randint(0, ad, (37, 1)).astype(np.float64) if disc else rng.randn(37, ad) * 2
        cost = ref.cost_function(obs, acs)
        orc = o_nets.CostNet(od, ad, hid, disc, None, None, 20, kw["action_low"], kw["action_high"])
        orc.load_state_dict(ref.network.state_dict())
        assert orc.select_dim == ref.select_dim
        d = maxdiff(orc.cost_function(obs, acs), cost); assert d == 0.0, d
        out.update({f"{name}/obs": obs, f"{name}/acs": acs, f"{name}/cost": cost,
                    f"{name}/select_dim": np.array(ref.select_dim), f"{name}/hidden": np.array(hid)})
        out.update(_sd_np(ref.network.state_dict(), f"{name}/w/"))
    # --cn_normalize (constraint_net.py:275-299): observations standardised with the given mean / var (eps 1e-5) BEFORE clipping
    th.manual_seed(3)
    mean, var = rng.randn(18) * 2, rng.rand(18) * 4 + 0.05
    kw = _cn_kwargs(18, 6, [20], None, None, initial_obs_mean=mean, initial_obs_var=var)
    ref = ConstraintNet(**kw)
    obs, acs = rng.randn(41, 18) * 12 + 1, rng.randn(41, 6) * 2
    cost = ref.cost_function(obs, acs)
    orc = o_nets.CostNet(18, 6, [20], False, None, None, 20, kw["action_low"], kw["action_high"])
    orc.load_state_dict(ref.network.state_dict()); orc.obs_mean, orc.obs_var = mean, var
    d = maxdiff(orc.cost_function(obs, acs), cost); assert d == 0.0, d
    out.update({"hc_norm/obs": obs, "hc_norm/acs": acs, "hc_norm/cost": cost, "hc_norm/mean": mean, "hc_norm/var": var,
                "hc_norm/select_dim": np.array(ref.select_dim), "hc_norm/hidden": np.array([20])})
    out.update(_sd_np(ref.network.state_dict(), "hc_norm/w/"))
    # the committed transfer checkpoint through the reference's own (positionally shifted) load()
    path = f"{REF}/icrl/expert_data/ConstraintTransfer/ICRL/AntBroken/files/best_cn_model.pt"
    raw = th.load(path)
    loaded = ConstraintNet.load(path)
    obs, acs = rng.randn(29, 113) * 30, rng.randn(29, 8) * 3
    cost = loaded.cost_function(obs, acs)
    print("  loaded net attrs: clip_obs", loaded.clip_obs, "action_low", type(loaded.action_low).__name__,
          "action_high", loaded.action_high, "obs_mean", loaded.current_obs_mean)
    orc = o_nets.CostNet(113, 8, raw["hidden_sizes"], False, raw["obs_select_dim"], raw["acs_select_dim"],
                         clip_obs=None, action_low=None, action_high=None)   # what load() really builds
    orc.load_state_dict(raw["cn_network"])
    d = maxdiff(orc.cost_function(obs, acs), cost); assert d == 0.0, d
    out.update({"antbroken/obs": obs, "antbroken/acs": acs, "antbroken/cost": cost,
                "antbroken/hidden": np.array(raw["hidden_sizes"]), "antbroken/select_dim": np.array(loaded.select_dim)})
    out.update(_sd_np(raw["cn_network"], "antbroken/w/"))
    print("  oracle == reference bit-for-bit (fresh nets and ConstraintNet.load quirk)")
    save("g2_cost_function", **out)


class _ReplayVecEnv(VecEnv):
    """Feeds a pre-generated (obs, rew, done) stream through the reference's wrappers."""

    def __init__(self, obs, rew, done, reset_obs):
        self.o, self.r, self.d, self.ro, self.t = obs, rew, done, reset_obs, 0
        super().__init__(obs.shape[1], gym.spaces.Box(-np.inf, np.inf, (obs.shape[2],), np.float64),
                         gym.spaces.Box(-1, 1, (2,), np.float32))

    def reset(self): return self.ro.copy()
    def step_async(self, a): pass
    def step_wait(self):
        t = self.t; self.t += 1
        return self.o[t].copy(), self.r[t].copy(), self.d[t].copy(), [{} for _ in range(self.num_envs)]
    def close(self): pass
    def seed(self, seed=None): return [seed]
    def get_attr(self, *a, **k): return []
    def set_attr(self, *a, **k): pass
    def env_method(self, *a, **k): return []


def g3_vecnormalize():
    print("G3 VecNormalizeWithCost stream")
    rng = np.random.RandomState(4)
    S, N, D = 50, 6, 5
    obs = rng.randn(S, N, D) * np.array([1, 5, 0.1, 30, 2.0]) + np.array([0, 3, -1, 10, 0.5])
    rew, done = rng.randn(S, N) * 4, rng.rand(S, N) < 0.1
    costs = rng.rand(S, N).astype(np.float32)
    reset_obs = rng.randn(N, D)
    venv = VecCostWrapper(_ReplayVecEnv(obs, rew, done, reset_obs))
    step_idx = {"t": 0}
    venv.set_cost_function(lambda o, a: costs[step_idx["t"]])
    env = VecNormalizeWithCost(venv, training=True, norm_obs=True, norm_reward=True, norm_cost=True,
                               cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.97)
    st = o_stats.NormState(N, D, reward_gamma=0.99, cost_gamma=0.97)
    rec = {k: [] for k in ("obs_n", "rew_n", "cost_n", "obs_mean", "obs_var", "obs_count", "ret_var", "ret_count",
                           "cost_var", "cost_count", "ret_mean", "cost_mean")}
    o0 = env.reset(); o0_o = o_stats.norm_reset(st, reset_obs)
    worst = maxdiff(o0, o0_o)
    for t in range(S):
        step_idx["t"] = t
        o, r, d, infos = env.step(np.zeros((N, 2), np.float32))
        c = np.array([i["cost"] for i in infos])
        oo, ro, co = o_stats.norm_step(st, obs[t], rew[t], costs[t], done[t])
        worst = max(worst, maxdiff(o, oo), maxdiff(r, ro), maxdiff(c, co), maxdiff(env.obs_rms.var, st.obs_rms.var),
                    maxdiff(env.cost_rms.var, st.cost_rms.var), maxdiff(env.ret_rms.mean, st.ret_rms.mean))
        assert np.array_equal(env.get_original_cost(), costs[t])
        for k, v in (("obs_n", o), ("rew_n", r), ("cost_n", c), ("obs_mean", env.obs_rms.mean), ("obs_var", env.obs_rms.var),
                     ("obs_count", env.obs_rms.count), ("ret_var", env.ret_rms.var), ("ret_count", env.ret_rms.count),
                     ("cost_var", env.cost_rms.var), ("cost_count", env.cost_rms.count), ("ret_mean", env.ret_rms.mean),
                     ("cost_mean", env.cost_rms.mean)):
            rec[k].append(np.array(v, copy=True))
    assert worst == 0.0, worst
    print("  oracle == reference bit-for-bit over", S, "steps")
    save("g3_vecnormalize", obs=obs, rew=rew, done=done, costs=costs, reset_obs=reset_obs, reset_obs_n=o0,
         gammas=np.array([0.99, 0.97]), **{k: np.array(v) for k, v in rec.items()})


def _make_ref_agent(n_envs, kind, seed, cn_hidden, **kw):
    env = RefSynthVecEnv(n_envs, kind, seed)
    env = VecCostWrapper(env)
    env = VecNormalizeWithCost(env, training=True, norm_obs=True, norm_reward=True, norm_cost=True,
                               cost_info_str="cost", reward_gamma=0.99, cost_gamma=0.99)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    th.manual_seed(100 + seed)
    cn = ConstraintNet(**_cn_kwargs(od, ad, cn_hidden, None, None, per_step_importance_sampling=True))
    env.set_cost_function(cn.cost_function)
    args = dict(n_steps=32, batch_size=16, n_epochs=3, target_kl=0.01, penalty_initial_value=1, penalty_learning_rate=0.1,
                budget=0.0, seed=seed, device="cpu", verbose=0,
                policy_kwargs=dict(net_arch=[dict(pi=[64, 64], vf=[64, 64], cvf=[64, 64])]))
    args.update(kw)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, **args)
    return agent, env, cn


def g13_widths():
    """g4 with network widths other than 64 and different per branch (-pl / -rvl / -cvl, icrl/utils.py:636-655)."""
    g4_ppo_minibatch("g13_widths", dict(pi=[32, 48], vf=[64, 32], cvf=[16, 64]))


def g15_wide():
    """g4 with layers wider than 64 (the generic-shape path of the build: csrc/generic.hip), different per branch."""
    g4_ppo_minibatch("g15_wide", dict(pi=[128, 128], vf=[96, 128], cvf=[128, 80]))


def g17_trunk():
    """g4 with a shared trunk and branches of 3 / 1 / 0 layers (-sl 48 -pl 64 32 32 -rvl 40 -cvl; torch_layers.py:129-254): the
    architectures icrl_policy_t.arch describes (csrc/generic.hip)."""
    g4_ppo_minibatch("g17_trunk", [48, dict(pi=[64, 32, 32], vf=[40], cvf=[])])


def g18_deep():
    """g4 with branches of 1 / 3 / 4 layers, no trunk (-pl 32 -rvl 64 64 64 -cvl 96 200 64 16)."""
    g4_ppo_minibatch("g18_deep", [dict(pi=[32], vf=[64, 64, 64], cvf=[96, 200, 64, 16])])


def g16_batch512():
    """g4 on ONE 512-row batch (batch sizes above 256: generic-shape path)."""
    g4_ppo_minibatch("g16_batch512", None, B=512)


def g14_batch256():
    """a 256-row minibatch (four 64-row chunks of the update kernels) through the reference's own policy / optimizer objects."""
    g4_ppo_minibatch("g14_batch256", None, B=256)


def g4_ppo_minibatch(name="g4_ppo_minibatch", arch=None, B=64):
    print("G4 PPO-Lagrangian minibatch step + train()" + ("" if arch is None else f" {arch}") + ("" if B == 64 else f" batch {B}"))
    net_arch = None if arch is None else ([dict(arch)] if isinstance(arch, dict) else list(arch))      # arch: the dict, or a full net_arch list
    kw = {} if arch is None else dict(policy_kwargs=dict(net_arch=net_arch))
    agent, env, cn = _make_ref_agent(4, "hc", 0, [20], **kw)
    pol = agent.policy
    sd0 = _sd_np(pol.state_dict())
    rng = np.random.RandomState(5 if B == 64 else 500 + B)
    obs = th.tensor(rng.randn(B, 18), dtype=th.float32); act = th.tensor(rng.randn(B, 6), dtype=th.float32)
    old_lp = th.tensor(-8 + rng.randn(B) * 0.3, dtype=th.float32)
    adv_r, adv_c = th.tensor(rng.randn(B), dtype=th.float32), th.tensor(rng.rand(B), dtype=th.float32)
    ret_r, ret_c = th.tensor(rng.randn(B), dtype=th.float32), th.tensor(rng.randn(B), dtype=th.float32)
    nu, clip = 0.731, 0.2
    # oracle policy with the same weights
    n_sh = 0 if net_arch is None else next(i for i, x in enumerate(net_arch) if isinstance(x, dict))
    hidden = (64, 64) if arch is None else dict(policy_net=net_arch[n_sh]["pi"], value_net=net_arch[n_sh]["vf"], cost_value_net=net_arch[n_sh]["cvf"])
    op = o_nets.TwoCriticPolicy(18, 6, hidden=hidden, shared=() if net_arch is None else tuple(net_arch[:n_sh]))
    assert list(op.params) == list(pol.state_dict())
    op.load_state_dict(pol.state_dict())
    oopt = th.optim.Adam(op.parameters(), lr=3e-4, eps=1e-5)
    out = {}
    for step in range(3):                                # three consecutive steps on the same batch (Adam state evolves)
        # --- reference arithmetic, invoked through the reference's own policy/optimizer objects
        v_r, v_c, lp, ent = pol.evaluate_actions(obs, act)
        a_r = adv_r - adv_r.mean(); a_r = a_r / (adv_r.std() + 1e-8); a_c = adv_c - adv_c.mean()
        ratio = th.exp(lp - old_lp)
        pl = -th.min(a_r * ratio, a_r * th.clamp(ratio, 1 - clip, 1 + clip)).mean()
        pl = (pl + nu * th.mean(a_c * ratio)) / (1 + nu)
        rvl = th.nn.functional.mse_loss(ret_r, v_r.flatten()); cvl = th.nn.functional.mse_loss(ret_c, v_c.flatten())
        loss = pl + 0.0 * (-th.mean(ent)) + 0.5 * rvl + 0.5 * cvl
        pol.optimizer.zero_grad(); loss.backward()
        grads = {k: p.grad.detach().numpy().copy() for k, p in pol.named_parameters()}
        gn = th.nn.utils.clip_grad_norm_(pol.parameters(), 0.5)
        pol.optimizer.step()
        # --- oracle
        oloss, tr = o_ppo.minibatch_loss(op, obs, act, old_lp, adv_r, adv_c, ret_r, ret_c, ret_r, ret_c, nu, clip)
        oopt.zero_grad(); oloss.backward()
        for k, p in op.params.items():
            assert maxdiff(p.grad.numpy(), grads[k]) == 0.0, k
        tot, coef = o_ppo.clip_coef_explicit([p.grad for p in op.parameters()], 0.5)
        assert abs(tot - float(gn)) < 1e-5 * max(1, tot)
        th.nn.utils.clip_grad_norm_(op.parameters(), 0.5); oopt.step()
        for k, p in op.params.items():
            assert maxdiff(p.detach().numpy(), pol.state_dict()[k].numpy()) == 0.0, k
        out.update({f"s{step}/loss": loss.item(), f"s{step}/policy_loss": pl.item(), f"s{step}/rvl": rvl.item(),
                    f"s{step}/cvl": cvl.item(), f"s{step}/entropy_loss": (-th.mean(ent)).item(),
                    f"s{step}/approx_kl": th.mean(old_lp - lp).item(),
                    f"s{step}/clip_fraction": th.mean((th.abs(ratio - 1) > clip).float()).item(),
                    f"s{step}/grad_norm": float(gn), f"s{step}/log_prob": lp.detach().numpy(),
                    f"s{step}/v_r": v_r.detach().numpy().ravel(), f"s{step}/v_c": v_c.detach().numpy().ravel()})
        out.update({f"s{step}/grad/{k}": v for k, v in grads.items()})
        out.update(_sd_np(pol.state_dict(), f"s{step}/after/"))
    print("  oracle == reference bit-for-bit (loss terms, grads, params after 3 Adam steps)")
    save(name, obs=obs.numpy(), act=act.numpy(), old_lp=old_lp.numpy(), adv_r=adv_r.numpy(),
         adv_c=adv_c.numpy(), ret_r=ret_r.numpy(), ret_c=ret_c.numpy(), nu=nu, clip=clip, lr=3e-4,
         **{f"w0/{k}": v for k, v in sd0.items()}, **out)


def g5_dual():
    print("G5 DualVariable trajectories")
    rng = np.random.RandomState(6)
    out = {}
    for name, (nu0, lr, budget) in dict(a=(1.0, 0.1, 0.0), b=(0.1, 0.05, 0.0), c=(1.0, 1.0, 0.0), d=(0.1, 1.0, 0.02)).items():
        costs = np.concatenate([rng.rand(60) * 0.5, np.zeros(90), rng.rand(50) * 0.05]).astype(np.float32)
        d = DualVariable(budget, lr, nu0, None)
        od = o_ppo.Dual(budget, lr, nu0, None)
        traj, worst = [], 0.0
        for c in costs:
            d.update_parameter(c); od.update(c)
            traj.append([d.nu().item(), d.loss.item(), d.nu.log_nu.item()])
            worst = max(worst, abs(d.nu().item() - od.nu().item()))
        assert worst == 0.0
        out[f"{name}/costs"], out[f"{name}/traj"], out[f"{name}/params"] = costs, np.array(traj), np.array([nu0, lr, budget])
        print(f"  {name}: nu0={nu0} lr={lr} -> min nu {np.min(np.array(traj)[:, 0]):.6g} (clamp floor)")
    save("g5_dual", **out)


def g11_pid():
    """PIDLagrangian (cpg --use_pid; dual_variable.py:60-122): 3 parameter sets x 200 costs."""
    print("G11 PIDLagrangian trajectories")
    from stable_baselines3.common.dual_variable import PIDLagrangian
    rng = np.random.RandomState(12)
    out = {}
    sets = dict(default_cpg=dict(alpha=0.0, penalty_init=1.0, Kp=10, Ki=0.0001, Kd=0, pid_delay=1, delta_p_ema_alpha=0.5, delta_d_ema_alpha=0.5),
                derivative=dict(alpha=0.02, penalty_init=0.1, Kp=1.0, Ki=0.01, Kd=5.0, pid_delay=10, delta_p_ema_alpha=0.95, delta_d_ema_alpha=0.9),
                integral=dict(alpha=0.1, penalty_init=0.0, Kp=0.0, Ki=0.5, Kd=0.0, pid_delay=3, delta_p_ema_alpha=0.0, delta_d_ema_alpha=0.0))
    for name, kw in sets.items():
        costs = np.concatenate([rng.rand(50) * 0.3, np.zeros(40), np.linspace(0, 0.5, 60), rng.rand(50) * 0.05]).astype(np.float32)
        ref, orc = PIDLagrangian(**kw), o_ppo.PID(**kw)
        traj = []
        for c in costs:
            ref.update_parameter(c); orc.update(c)
            assert ref.nu().item() == orc.nu().item() and ref.loss.item() == orc.loss.item()
            traj.append([ref.nu().item(), ref.loss.item(), ref.pid_i, ref._delta_p, ref._cost_delta])
        out[f"{name}/costs"], out[f"{name}/traj"] = costs, np.array(traj)
        out[f"{name}/params"] = np.array([kw[k] for k in ("alpha", "penalty_init", "Kp", "Ki", "Kd", "pid_delay", "delta_p_ema_alpha", "delta_d_ema_alpha")])
        print(f"  {name}: nu range [{np.min(np.array(traj)[:, 0]):.4g}, {np.max(np.array(traj)[:, 0]):.4g}]")
    print("  oracle == reference bit-for-bit")
    save("g11_pid", **out)


def g6_constraint_net_train():
    print("G6 compute_is_weights + ConstraintNet.train")
    rng = np.random.RandomState(7)
    out = {}
    cases = dict(psis=(True, [30, 50, 20], 6, 10, 2.5), episode=(False, [10, 20, 70], 6, 10, 10),
                 overflow=(True, [600, 400], 8, 10, 2.5), earlystop=(False, [40, 60], 12, 0.001, 0.0005))
    for name, (psis, lengths, iters, tk_on, tk_no) in cases.items():
        n_nom, n_exp = int(np.sum(lengths)), 150
        exp_obs, exp_acs = rng.randn(n_exp, 18), rng.uniform(-1, 1, (n_exp, 6)).astype(np.float32)
        nom_obs, nom_acs = rng.randn(n_nom, 18) * (3 if name == "overflow" else 1), rng.uniform(-1.5, 1.5, (n_nom, 6))
        th.manual_seed(11)
        ref = ConstraintNet(**_cn_kwargs(18, 6, [20], exp_obs, exp_acs, per_step_importance_sampling=psis,
                                         target_kl_old_new=tk_on, target_kl_new_old=tk_no,
                                         lr_schedule=(lambda x: 0.2) if name == "overflow" else (lambda x: 0.05)))
        w0 = _sd_np(ref.network.state_dict())
        orc = o_nets.CostNet(18, 6, [20], False, None, None, 20, ref.action_low, ref.action_high)
        orc.load_state_dict(ref.network.state_dict())
        oopt = th.optim.Adam(orc.parameters(), lr=ref.lr_schedule(1), eps=1e-5)
        m = ref.train(iters, nom_obs, nom_acs, np.array(lengths))
        om = o_cn.cn_train(orc, oopt, iters, orc.prepare(nom_obs, nom_acs), orc.prepare(exp_obs, exp_acs), np.array(lengths),
                           reg_coeff=0.5, per_step=psis, target_kl_old_new=tk_on, target_kl_new_old=tk_no)
        for k, v in m.items():
            a, b = float(v), float(om[k])
            assert (np.isnan(a) and np.isnan(b)) or a == b, (name, k, a, b)
        for k, p in orc.params.items():
            assert maxdiff(p.detach().numpy(), ref.network.state_dict()[k].numpy()) == 0.0
        print(f"  {name}: early_stop_itr={m['backward/early_stop_itr']} kl_on={m['backward/kl_old_new']:.4g} kl_no={m['backward/kl_new_old']:.4g}")
        out.update({f"{name}/exp_obs": exp_obs, f"{name}/exp_acs": exp_acs, f"{name}/nom_obs": nom_obs, f"{name}/nom_acs": nom_acs,
                    f"{name}/lengths": np.array(lengths), f"{name}/cfg": np.array([psis, iters, tk_on, tk_no, ref.lr_schedule(1)], np.float64)})
        out.update({f"{name}/w0/{k}": v for k, v in w0.items()})
        out.update(_sd_np(ref.network.state_dict(), f"{name}/w1/"))
        out.update({f"{name}/m/{k.split('/')[1]}": float(v) for k, v in m.items()})
    # stand-alone compute_is_weights
    ref = ConstraintNet(**_cn_kwargs(18, 6, [20], None, None, per_step_importance_sampling=False))
    po, pn = th.tensor(rng.rand(30, 1), dtype=th.float32), th.tensor(rng.rand(30, 1), dtype=th.float32)
    for psis in (False, True):
        ref.per_step_importance_sampling = psis
        w, a, b = ref.compute_is_weights(po.clone(), pn.clone(), np.array([10, 20]))
        ow, oa, ob = o_cn.is_weights_and_kls(po.clone(), pn.clone(), np.array([10, 20]), 1e-5, psis)
        assert w.shape == ow.shape and maxdiff(w, ow) == 0 and a == oa and b == ob
        out.update({f"isw{int(psis)}/w": w.numpy(), f"isw{int(psis)}/kl": np.array([a.item(), b.item()])})
    out.update({"isw/po": po.numpy(), "isw/pn": pn.numpy()})
    print("  oracle == reference bit-for-bit (metrics, weights; inf/nan case included)")
    save("g6_constraint_net", **out)


def g7_constraint_net_minibatch():
    """ConstraintNet.train with cn_batch_size (constraint_net.py:181-206, get() :300-316): permutations recorded."""
    print("G7 ConstraintNet.train, minibatch mode")
    rng = np.random.RandomState(17)
    np.random.seed(29)          # get() draws its permutations from the global generator (constraint_net.py:300-316)
    out = {}
    cases = dict(mb_psis=(True, [60, 90, 50], 4, -1, 200.0, 32, False), mb_episode=(False, [80, 70, 30], 3, 10, 10, 48, False),
                 mb_nois=(False, [100, 60], 3, -1, -1, 64, True), mb_gail=(False, [70, 80], 3, -1, -1, 40, "gail"))
    for name, (psis, lengths, iters, tk_on, tk_no, bs, mode) in cases.items():
        n_nom, n_exp = int(np.sum(lengths)), 170
        exp_obs, exp_acs = rng.randn(n_exp, 18), rng.uniform(-1, 1, (n_exp, 6)).astype(np.float32)
        nom_obs, nom_acs = rng.randn(n_nom, 18), rng.uniform(-1.5, 1.5, (n_nom, 6))
        th.manual_seed(13)
        nois, gail = mode is True, mode == "gail"
        ref = ConstraintNet(**_cn_kwargs(18, 6, [20], exp_obs, exp_acs, per_step_importance_sampling=psis, batch_size=bs,
                                         target_kl_old_new=tk_on, target_kl_new_old=tk_no, no_importance_sampling=nois or gail,
                                         train_gail_lambda=gail))
        w0 = _sd_np(ref.network.state_dict())
        orc = o_nets.CostNet(18, 6, [20], False, None, None, 20, ref.action_low, ref.action_high)
        orc.load_state_dict(ref.network.state_dict())
        oopt = th.optim.Adam(orc.parameters(), lr=ref.lr_schedule(1), eps=1e-5)
        perms, orig_perm = [], np.random.permutation

        def rec_perm(n):
            p = orig_perm(n); perms.append(p.copy()); return p
        np.random.permutation = rec_perm
        try:
            m = ref.train(iters, nom_obs, nom_acs, np.array(lengths))
        finally:
            np.random.permutation = orig_perm
        replay = iter(perms)
        om = o_cn.cn_train(orc, oopt, iters, orc.prepare(nom_obs, nom_acs), orc.prepare(exp_obs, exp_acs), np.array(lengths),
                           reg_coeff=0.5, per_step=psis, target_kl_old_new=tk_on, target_kl_new_old=tk_no, batch_size=bs,
                           importance_sampling=not (nois or gail), gail=gail,
                           rng=type("R", (), {"permutation": staticmethod(lambda n: next(replay))}))
        for k, v in m.items():
            a, b = float(v), float(om[k])
            assert (np.isnan(a) and np.isnan(b)) or a == b, (name, k, a, b)
        for k, p in orc.params.items():
            assert maxdiff(p.detach().numpy(), ref.network.state_dict()[k].numpy()) == 0.0
        print(f"  {name}: {len(perms)} permutations, loss {m['backward/cn_loss']:.6f}")
        out.update({f"{name}/exp_obs": exp_obs, f"{name}/exp_acs": exp_acs, f"{name}/nom_obs": nom_obs, f"{name}/nom_acs": nom_acs,
                    f"{name}/lengths": np.array(lengths), f"{name}/perms": np.array(perms),
                    f"{name}/cfg": np.array([psis, iters, tk_on, tk_no, ref.lr_schedule(1), bs, nois or gail, gail], np.float64)})
        out.update({f"{name}/w0/{k}": v for k, v in w0.items()})
        out.update(_sd_np(ref.network.state_dict(), f"{name}/w1/"))
        out.update({f"{name}/m/{k.split('/')[1]}": float(v) for k, v in m.items()})
    print("  oracle == reference bit-for-bit")
    save("g7_constraint_net_minibatch", **out)


G8_ARGV = ["icrl", "-er", "20", "-tei", "LGW-v0", "-eei", "CLGW-v0", "-tk", "0.01", "-cl", "20", "-clr", "0.003", "-ft", "800", "-ni", "3",
           "-bi", "20", "-dno", "-dnr", "-dnc", "--n_steps", "200", "-nt", "2", "-s", "7", "-d", "cpu"]


def _lgw_expert_dir(tmp):
    """the reference ships no LGW expert rollouts: write the committed fixture in the reference's on-disk layout
    (files/EXPERT/rollouts/{i}.pkl, icrl/icrl.py:25-43) next to the reference's own expert agent archive."""
    d = np.load(os.path.join(OUT, "expert_lgw.npz"))
    ep = os.path.join(tmp, "LGW")
    os.makedirs(os.path.join(ep, "files/EXPERT/rollouts"))
    os.symlink(f"{REF}/icrl/expert_data/LGW/files/best_model.zip", os.path.join(ep, "files/best_model.zip"))
    off = 0
    for i, L in enumerate(d["lengths"]):
        L = int(L)
        with open(os.path.join(ep, f"files/EXPERT/rollouts/{i}.pkl"), "wb") as f:
            pickle.dump(dict(observations=d["observations"][off:off + L], actions=d["actions"][off:off + L],
                             rewards=np.array([d["rewards"][i]]), lengths=np.array([L]), save_scheme="not_airl"), f)
        off += L
    return ep, d


G8_WALL_CLOCK = ("time(m)", "time/fps", "time/time_elapsed")      # what no second run can reproduce


def g8_icrl_lgw():
    """configs[0] end to end: the reference's OWN icrl(config) (icrl/icrl.py:45-312; real SubprocVecEnv workers, Monitor,
    plotting) on LGW-v0 / CLGW-v0 with the README.md:25 flags at a reduced size, 3 outer iterations.  Every random draw is
    recorded (Categorical.sample, np.random.permutation) together with the initial weights and the per-iteration metrics;
    the CPU port, teacher-forced with those draws, must reproduce the metrics."""
    print("G8 reference icrl() on LGW-v0, 3 outer iterations")
    import tempfile
    import types
    import wandb
    import icrl.icrl as ref_icrl
    from icrl_amd.icrl import build_parser          # same flag names / defaults as icrl/icrl.py:316-417 (host code, no GPU use)
    from oracle.streams import RecordedStreams
    tmp = tempfile.mkdtemp(prefix="g8_")
    ep, expert = _lgw_expert_dir(tmp)
    cfg = vars(build_parser().parse_args(G8_ARGV + ["-ep", ep]))
    cfg.update(save_dir=os.path.join(tmp, "run"), wandb_sweep=True)       # wandb_sweep: skip the final video (icrl.py:307-309)
    os.makedirs(cfg["save_dir"])
    # ---- recorders
    Cat = th.distributions.Categorical
    orig_sample, orig_perm = Cat.sample, np.random.permutation
    phase = ["learn"]
    draws = dict(learn=[], sample=[], eval=[])
    perms, snaps, logs = [], {}, []

    def rec_sample(self, sample_shape=th.Size()):
        a = orig_sample(self, sample_shape); draws[phase[0]].append(a.numpy().copy()); return a

    def rec_perm(n):
        p = orig_perm(n); perms.append(p.copy()); return p

    class RecPPO(ref_icrl.PPOLagrangian):
        def _setup_model(self):
            super()._setup_model()
            if "policy" not in snaps and self.env is not None:           # the nominal agent (the expert agent has env=None)
                snaps["policy"] = _sd_np(self.policy.state_dict()); snaps["agent"] = self

    class RecCN(ref_icrl.ConstraintNet):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            snaps["cn"] = _sd_np(self.network.state_dict()); snaps["cn_obj"] = self

    def phased(name, fn, counts):
        def wrapped(*a, **k):
            phase[0] = name; n0 = len(draws[name])
            try:
                return fn(*a, **k)
            finally:
                phase[0] = "learn"; counts.append(len(draws[name]) - n0)
        return wrapped
    sample_counts, eval_counts = [], []
    saved = (ref_icrl.PPOLagrangian, ref_icrl.ConstraintNet, ref_icrl.utils.sample_from_agent, ref_icrl.evaluate_policy, wandb.log)
    ref_icrl.PPOLagrangian, ref_icrl.ConstraintNet = RecPPO, RecCN
    ref_icrl.utils.sample_from_agent = phased("sample", saved[2], sample_counts)
    ref_icrl.evaluate_policy = phased("eval", saved[3], eval_counts)
    wandb.log = lambda m: logs.append({k: float(v) for k, v in m.items() if np.ndim(v) == 0 and not isinstance(v, str)})
    Cat.sample, np.random.permutation = rec_sample, rec_perm
    try:
        ref_icrl.icrl(types.SimpleNamespace(**cfg))
    finally:
        Cat.sample, np.random.permutation = orig_sample, orig_perm
        (ref_icrl.PPOLagrangian, ref_icrl.ConstraintNet, ref_icrl.utils.sample_from_agent, ref_icrl.evaluate_policy, wandb.log) = saved
    T, N, ni = cfg["n_steps"], cfg["num_threads"], cfg["n_iters"]
    learn_actions = np.array(draws["learn"]).reshape(-1, T, N)
    sample_actions = np.array(draws["sample"]).reshape(ni, -1)
    eval_actions = np.array(draws["eval"]).reshape(-1)
    assert sample_actions.shape[1] == cfg["expert_rollouts"] * 200 and len(eval_counts) == ni and sum(eval_counts) == len(eval_actions)
    print(f"  reference: {learn_actions.shape[0]} rollouts, {len(perms)} permutations, eval steps per iteration {eval_counts}")
    print("  nu", [round(m["forward/nu"], 6) for m in logs], "true/cost", [m["true/cost"] for m in logs])
    g = dict(learn_actions=learn_actions.astype(np.int8), perms=np.array(perms).astype(np.int16), sample_actions=sample_actions.astype(np.int8),
             eval_actions=eval_actions.astype(np.int8), eval_counts=np.array(eval_counts))
    # ---- the CPU port, teacher-forced
    esd = th.load(io.BytesIO(zipfile.ZipFile(os.path.join(ep, "files/best_model.zip")).read("policy.pth")))
    port_cfg = {k: cfg[k] for k in o_loop.PORT_DEFAULTS if k in cfg}
    om, _, _, objs = o_loop.icrl_port(port_cfg, expert["observations"][:4000], expert["actions"][:4000], esd, streams=RecordedStreams(g),
                                      init=dict(policy=snaps["policy"], cn=snaps["cn"]))
    # the golden file holds EVERY scalar the reference logged: the consumers decide what they skip (wall-clock keys,
    # Monitor's episode statistics) in an explicit list; the port is compared here on what it produces
    keys = sorted(logs[0])
    assert all(sorted(m) == keys for m in logs), "the reference logged different key sets in different iterations"
    cmp_keys = [k for k in keys if k in om[0] and k not in G8_WALL_CLOCK]
    worst = {}
    for it in range(ni):
        for k in cmp_keys:
            a, b = logs[it][k], float(om[it][k])
            worst[k] = max(worst.get(k, 0.0), 0.0 if (np.isnan(a) and np.isnan(b)) else abs(a - b) / max(1.0, abs(a)))
    missing = sorted(k for k in logs[0] if k not in om[0])
    print("  port vs reference, worst relative deviation per key (>0 only):", {k: float(f"{v:.3g}") for k, v in worst.items() if v > 0})
    print("  reference keys the port does not produce:", missing)
    assert max(worst.values()) < 2e-5, worst
    fin_p, fin_c = snaps["agent"].policy.state_dict(), snaps["cn_obj"].network.state_dict()
    dw = max(maxdiff(objs["agent"].policy.params[k].detach().numpy(), fin_p[k].numpy()) for k in fin_p)
    dc = max(maxdiff(objs["cn"].params[k].detach().numpy(), fin_c[k].numpy()) for k in fin_c)
    print(f"  final weights port vs reference: policy {dw:.3g}, constraint net {dc:.3g}")
    assert dw < 1e-5 and dc < 1e-5
    save("g8_icrl_lgw", argv=np.array(G8_ARGV), metric_keys=np.array(keys), metrics=np.array([[logs[it][k] for k in keys] for it in range(ni)]),
         **g, **{f"w0/{k}": v for k, v in snaps["policy"].items()}, **{f"cn0/{k}": v for k, v in snaps["cn"].items()},
         **{f"w1/{k}": v.numpy() for k, v in fin_p.items()}, **{f"cn1/{k}": v.numpy() for k, v in fin_c.items()},
         **{f"expert_policy/{k}": v.numpy() for k, v in esd.items()})


def g12_gail():
    """GAIL-constraint baseline (SURVEY f-4): the reference's GailDiscriminator + GailCallback + plain PPO on the synthetic env,
    2 rollouts, every draw recorded; the CPU port (oracle/gail.py on the two-critics PortAgent with zero costs) teacher-forced."""
    print("G12 GAIL discriminator + callback + PPO")
    import tempfile
    import torch.distributions.normal as tdn
    from stable_baselines3 import PPO
    from icrl.gail_utils import GailCallback, GailDiscriminator
    from oracle import gail as o_gail
    N, T = 4, 32
    ex = np.load(os.path.join(OUT, "expert_hc.npz"))
    exp_obs, exp_acs = ex["observations"][:300].astype(np.float64), ex["actions"][:300]
    env = VecNormalizeWithCost(RefSynthVecEnv(N, "hc", 0), training=True, norm_obs=True, norm_reward=True, norm_cost=False, reward_gamma=0.99)
    th.manual_seed(5)
    disc = GailDiscriminator(18, 6, [20], 48, lambda x: 0.01, exp_obs, exp_acs, False, clip_obs=20, eps=1e-5)
    d0 = _sd_np(disc.network.state_dict())
    agent = PPO("MlpPolicy", env, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.02, seed=0, device="cpu", verbose=0,
                policy_kwargs=dict(net_arch=[dict(pi=[64, 64], vf=[64, 64])]))
    sd0 = _sd_np(agent.policy.state_dict())
    wall = lambda o, a: (o[..., 0] <= -0.05)
    cb = GailCallback(disc, False, wall, tempfile.mkdtemp(prefix="g12_"), plot_disc=False)
    noise, perms, tags, phase, disc_metrics, relabelled = [], [], [], ["ppo"], [], []
    orig_sn, orig_perm, orig_train, orig_end = tdn._standard_normal, np.random.permutation, disc.train, cb._on_rollout_end

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype, device); noise.append(e.numpy().copy()); return e

    def rec_perm(n):
        p = orig_perm(n); perms.append(p.copy()); tags.append(phase[0]); return p

    def rec_train(*a, **k):
        phase[0] = "disc"
        try:
            m = orig_train(*a, **k); disc_metrics.append(dict(m)); return m
        finally:
            phase[0] = "ppo"

    def rec_end():
        orig_end()
        relabelled.append((agent.rollout_buffer.rewards.copy(), agent.rollout_buffer.advantages.copy(), agent.rollout_buffer.returns.copy(),
                           float(ref_logger.Logger.CURRENT.name_to_value["eval/mean_cost"])))
    tdn._standard_normal, np.random.permutation, disc.train, cb._on_rollout_end = rec_sn, rec_perm, rec_train, rec_end
    try:
        agent.learn(total_timesteps=2 * N * T, callback=cb)
    finally:
        tdn._standard_normal, np.random.permutation = orig_sn, orig_perm
    logs = {k: float(v) for k, v in ref_logger.Logger.CURRENT.name_to_value.items() if k.startswith("train/")}
    noise = np.array(noise).reshape(2, T, N, 6)
    print("  reference: permutations", tags, "disc loss", [round(m["discriminator/disc_loss"], 5) for m in disc_metrics],
          "mean cost", [r[3] for r in relabelled])
    # ---- CPU port, teacher-forced
    stack = o_loop.make_stack(N, "hc", 0, norm_cost=False)
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.02, seed=0, cost_vf_coef=0.0, penalty_initial_value=0.0)
    psd = {k: v.clone() for k, v in port.policy.state_dict().items()}
    for k, v in sd0.items():
        psd[k] = th.as_tensor(v)
    port.policy.load_state_dict(psd)
    onet = o_gail.make_disc(18, 6, [20])
    onet.load_state_dict({k: th.as_tensor(v) for k, v in d0.items()})
    oopt = th.optim.Adam(onet.parameters(), lr=0.01, eps=1e-5)
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    cursor, worst = 0, {}
    for it in range(2):
        port.collect_rollouts(noise[it])
        assert tags[cursor] == "disc"
        replay = iter([perms[cursor]]); cursor += 1
        m = o_gail.rollout_end(port, onet, oopt, exp_obs, exp_acs, port.last_values, port.last_dones_out, batch_size=48,
                               true_cost_fn=wall, rng=type("R", (), {"permutation": staticmethod(lambda n: next(replay))}))
        for k, v in disc_metrics[it].items():
            worst["disc/" + k] = max(worst.get("disc/" + k, 0.0), abs(v - m[k]))
        worst["mean_cost"] = max(worst.get("mean_cost", 0.0), abs(relabelled[it][3] - m["eval/mean_cost"]))
        flat = lambda a: a.reshape(N, T).swapaxes(0, 1) if a.shape[0] == N * T else a.reshape(T, N)
        worst["rewards"] = max(worst.get("rewards", 0.0), maxdiff(relabelled[it][0].reshape(T, N), port.buf.rewards))
        worst["advantages"] = max(worst.get("advantages", 0.0), maxdiff(relabelled[it][1].reshape(T, N), port.buf.reward_advantages))
        remaining = perms[cursor:]
        res = port.train(lambda e, r=remaining: r[e])
        used = min(int(res["train/early_stop_epoch"]) + 1, 3)
        assert all(t == "ppo" for t in tags[cursor:cursor + used])
        cursor += used
    assert cursor == len(perms), (cursor, len(perms))
    for k in sd0:
        worst["w/" + k] = maxdiff(port.policy.params[k].detach().numpy(), agent.policy.state_dict()[k].numpy())
    for k in d0:
        worst["d/" + k] = maxdiff(onet.params[k].detach().numpy(), disc.network.state_dict()[k].numpy())
    for a, b in (("train/approx_kl", "train/approx_kl"), ("train/value_loss", "train/reward_value_loss"), ("train/policy_gradient_loss", "train/policy_gradient_loss")):
        worst[a] = abs(logs[a] - float(res[b]))
    print("  port vs reference, max abs deviation:", {k: v for k, v in worst.items() if v > 0} or "all exactly 0")
    assert max(worst.values()) < 1e-6, worst
    rng = np.random.RandomState(3)
    po, pa = rng.randn(2, 5, 18) * 3, rng.uniform(-1, 1, (2, 5, 6)).astype(np.float32)
    save("g12_gail", noise=noise, perms=np.array(perms), perm_is_disc=np.array([t == "disc" for t in tags]), exp_obs=exp_obs, exp_acs=exp_acs,
         **{f"w0/{k}": v for k, v in sd0.items()}, **{f"d0/{k}": v for k, v in d0.items()},
         **{f"w1/{k}": v.numpy() for k, v in agent.policy.state_dict().items()}, **_sd_np(disc.network.state_dict(), "d1/"),
         **{f"disc_metrics/{it}/{k.split('/')[1]}": v for it, m in enumerate(disc_metrics) for k, v in m.items()},
         rewards=np.array([r[0].reshape(T, N) for r in relabelled]), advantages=np.array([r[1].reshape(T, N) for r in relabelled]),
         returns=np.array([r[2].reshape(T, N) for r in relabelled]), mean_cost=np.array([r[3] for r in relabelled]),
         probe_obs=po, probe_acs=pa, probe_reward=disc.reward_function(po, pa), probe_d=disc.reward_function(po, pa, apply_log=False),
         **{"log/" + k.split("/")[1]: v for k, v in logs.items()})


def g9_learn_iteration():
    """One learn() of the reference on the synthetic env (N=4, T=32) with the action noise and the minibatch
    permutations recorded, against the CPU port teacher-forced with the same streams."""
    print("G9 learn() iteration + sample_from_agent on the synthetic env")
    import torch.distributions.normal as tdn
    N, T = 4, 32
    agent, env, cn = _make_ref_agent(N, "hc", 0, [20])
    sd0 = _sd_np(agent.policy.state_dict()); cn0 = _sd_np(cn.network.state_dict())
    noise, perms = [], []
    orig_sn, orig_perm = tdn._standard_normal, np.random.permutation

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype, device); noise.append(e.numpy().copy()); return e

    def rec_perm(n):
        p = orig_perm(n); perms.append(p.copy()); return p
    tdn._standard_normal, np.random.permutation = rec_sn, rec_perm
    try:
        agent.learn(total_timesteps=2 * N * T, cost_function="cost")
    finally:
        tdn._standard_normal, np.random.permutation = orig_sn, orig_perm
    logs = {k: float(v) for k, v in ref_logger.Logger.CURRENT.name_to_value.items() if k.startswith("train/")}
    rb = agent.rollout_buffer
    noise = np.array(noise).reshape(2, T, N, 6)
    n_ep = [int(logs["train/early_stop_epoch"])]
    print("  reference: early_stop_epoch", logs["train/early_stop_epoch"], "nu", logs["train/nu"], "perms drawn", len(perms))
    # ---- CPU port, teacher-forced
    stack = o_loop.make_stack(N, "hc", 0)
    ocn = o_nets.CostNet(18, 6, [20], False, None, None, 20, -np.ones(6, np.float32), np.ones(6, np.float32))
    ocn.load_state_dict(cn.network.state_dict())
    stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, seed=0)
    port.policy.load_state_dict({k: th.as_tensor(v) for k, v in sd0.items()})
    # split the recorded permutations per rollout (the reference stops drawing after an early stop)
    per_iter, k = [], 0
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    worst = {}
    for it in range(2):
        b = port.collect_rollouts(noise[it])
        remaining = perms[k:]
        res = port.train(lambda e, r=remaining: r[e])
        used = res["train/early_stop_epoch"] + 1 if res["train/early_stop_epoch"] < 3 else 3
        per_iter.append(np.array(remaining[:used])); k += used
    assert k == len(perms), (k, len(perms))
    for key in ("observations", "orig_observations", "new_observations", "actions", "rewards", "costs", "orig_costs", "dones",
                "log_probs", "reward_values", "cost_values"):
        a = getattr(rb, key); a = a.reshape(N, T, -1).swapaxes(0, 1) if a.shape[0] == N * T else a
        worst[key] = maxdiff(a.reshape(T, N, -1), getattr(b, key).reshape(T, N, -1))
    for key in ("reward_advantages", "cost_advantages", "reward_returns", "cost_returns"):
        a = getattr(rb, key).reshape(N, T).swapaxes(0, 1)
        worst[key] = maxdiff(a, getattr(b, key))
    for key in ("train/nu", "train/average_cost", "train/approx_kl", "train/loss", "train/policy_gradient_loss", "train/std"):
        worst[key] = abs(logs[key] - float(res[key]))
    for kname, p in port.policy.params.items():
        worst["w/" + kname] = maxdiff(p.detach().numpy(), agent.policy.state_dict()[kname].numpy())
    print("  port vs reference, max abs deviation:", {k: v for k, v in worst.items() if v > 0} or "all exactly 0")
    assert max(worst.values()) < 1e-5, worst
    # ---- sample_from_agent pairing, reference vs port (1-env sampling env, synced stats)
    from stable_baselines3.common.vec_env import sync_envs_normalization
    senv = VecNormalizeWithCost(RefSynthVecEnv(1, "hc", 0), training=False, norm_obs=True, norm_reward=False, norm_cost=False)
    sync_envs_normalization(env, senv)
    snoise = []
    tdn._standard_normal = lambda shape, dtype, device: (lambda e: (snoise.append(e.numpy().copy()), e)[1])(orig_sn(shape, dtype, device))
    try:
        r_oo, r_o, r_a, r_r, r_l = ref_utils.sample_from_agent(agent, senv, 2)
    finally:
        tdn._standard_normal = orig_sn
    snoise = np.array(snoise).reshape(-1, 6)
    sstack = o_loop.make_stack(1, "hc", 0, training=False, norm_reward=False, norm_cost=False)
    o_loop.sync_normalization(stack.norm, sstack.norm)
    port.stack = sstack
    p_oo, p_o, p_a, p_r, p_l = o_loop.sample_from_agent(port, sstack, 2, snoise)
    port.stack = stack
    d = max(maxdiff(r_oo, p_oo), maxdiff(r_o, p_o), maxdiff(r_a, p_a), maxdiff(r_r, p_r)); assert np.array_equal(r_l, p_l)
    print("  sample_from_agent port vs reference max dev", d, "lengths", r_l)
    assert d < 1e-5
    flat = lambda a: a.reshape(N, T, -1).swapaxes(0, 1).reshape(T, N, -1) if a.shape[0] == N * T else a.reshape(T, N, -1)
    save("g9_learn_iteration", noise=noise, perms0=per_iter[0], perms1=per_iter[1], sample_noise=snoise,
         sample_orig_obs=r_oo, sample_obs=r_o, sample_actions=r_a, sample_rewards=r_r, sample_lengths=r_l,
         **{f"w0/{k}": v for k, v in sd0.items()}, **{f"cn/{k}": v for k, v in cn0.items()},
         **{f"w1/{k}": v.numpy() for k, v in agent.policy.state_dict().items()},
         **{f"buf/{k}": flat(getattr(rb, k)) for k in ("observations", "orig_observations", "new_observations", "actions", "rewards",
                                                      "costs", "orig_costs", "dones", "log_probs", "reward_values", "cost_values",
                                                      "reward_advantages", "cost_advantages", "reward_returns", "cost_returns")},
         **{"log/" + k.split("/")[1]: v for k, v in logs.items()},
         obs_rms_mean=env.obs_rms.mean, obs_rms_var=env.obs_rms.var, obs_rms_count=env.obs_rms.count,
         ret_rms_var=env.ret_rms.var, cost_rms_var=env.cost_rms.var)


def g10_lap_grid():
    """configs[0]: the reference's LapGridWorld / ConstrainedLapGridWorld stepped through gym.make + DummyVecEnv against the
    oracle restatement, then one learn() of the reference with a Categorical policy (actions and permutations recorded)
    against the CPU port teacher-forced with the same actions."""
    print("G10 LapGridWorld env + discrete-action learn()")
    import custom_envs  # noqa: F401  (registers LGW-v0 / CLGW-v0 with the gym stand-in)
    from stable_baselines3.common.vec_env import DummyVecEnv
    from oracle.lap_grid import LapGridVecEnv
    out = {}
    rng = np.random.RandomState(11)
    for env_id, constrained, p_back in (("LGW-v0", False, 0.3), ("CLGW-v0", True, 0.01)):
        N, S = 3, 700
        ref = DummyVecEnv([(lambda: gym.make(env_id)) for _ in range(N)])
        mine = LapGridVecEnv(N, constrained=constrained)
        acts = (rng.rand(S, N) < p_back).astype(np.int64)
        o_r, o_m = ref.reset(), mine.reset()
        assert maxdiff(o_r, o_m) == 0.0
        obs, rew, done = [], [], []
        for t in range(S):
            o_r, r_r, d_r, _ = ref.step(acts[t])
            o_m, r_m, d_m = mine.step(acts[t])
            assert maxdiff(o_r, o_m.astype(np.float32)) == 0.0 and maxdiff(r_r, r_m) == 0.0 and np.array_equal(d_r, d_m), (env_id, t)
            obs.append(o_m.copy()); rew.append(r_m.copy()); done.append(d_m.copy())
        k = env_id.split("-")[0].lower()
        out.update({f"{k}/actions": acts, f"{k}/obs": np.array(obs), f"{k}/rew": np.array(rew), f"{k}/done": np.array(done)})
        print(f"  {env_id}: oracle == reference over {S} steps x {N} envs ({int(np.sum(done))} episode ends)")
    # ---- learn() with the reference's classes on LGW-v0 (normalisation switched off as in README.md:25: -dno -dnr -dnc)
    N, T = 2, 64
    env = DummyVecEnv([(lambda: gym.make("LGW-v0")) for _ in range(N)])
    env = VecCostWrapper(env)
    env = VecNormalizeWithCost(env, training=True, norm_obs=False, norm_reward=False, norm_cost=False)
    th.manual_seed(321)
    cn = ConstraintNet(**_cn_kwargs(1, 2, [20], None, None, is_discrete=True))
    env.set_cost_function(cn.cost_function)
    agent = PPOLagrangian("TwoCriticsMlpPolicy", env, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, penalty_initial_value=1,
                          penalty_learning_rate=0.1, budget=0.0, seed=3, device="cpu", verbose=0, ent_coef=0.01,
                          policy_kwargs=dict(net_arch=[dict(pi=[64, 64], vf=[64, 64], cvf=[64, 64])]))
    sd0 = _sd_np(agent.policy.state_dict()); cn0 = _sd_np(cn.network.state_dict())
    actions, perms = [], []
    Cat = th.distributions.Categorical
    orig_sample, orig_perm = Cat.sample, np.random.permutation

    def rec_sample(self, sample_shape=th.Size()):
        a = orig_sample(self, sample_shape); actions.append(a.numpy().copy()); return a

    def rec_perm(n):
        p = orig_perm(n); perms.append(p.copy()); return p
    Cat.sample, np.random.permutation = rec_sample, rec_perm
    try:
        agent.learn(total_timesteps=2 * N * T, cost_function="cost")
    finally:
        Cat.sample, np.random.permutation = orig_sample, orig_perm
    logs = {k: float(v) for k, v in ref_logger.Logger.CURRENT.name_to_value.items() if k.startswith("train/")}
    rb = agent.rollout_buffer
    acts = np.array(actions).reshape(2, T, N)
    print("  reference: early_stop_epoch", logs["train/early_stop_epoch"], "nu", logs["train/nu"], "perms drawn", len(perms))
    stack = o_loop.make_stack(N, "lgw", 0, norm_obs=False, norm_reward=False, norm_cost=False)
    ocn = o_nets.CostNet(1, 2, [20], True, None, None, 20, None, None)
    ocn.load_state_dict(cn.network.state_dict())
    stack.cost_fn = ocn.cost_function
    port = o_loop.PortAgent(stack, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.01, seed=3, discrete=True, ent_coef=0.01)
    port.policy.load_state_dict({k: th.as_tensor(v) for k, v in sd0.items()})
    port.num_timesteps = 0
    port._last_obs = stack.reset(); port._last_dones = np.zeros(N, bool); port._last_original_obs = stack.old_obs.copy()
    per_iter, k, worst = [], 0, {}
    for it in range(2):
        b = port.collect_rollouts(acts[it].astype(np.float32))    # uniforms 0.0 / 1.0 force action 0 / 1 (two classes)
        remaining = perms[k:]
        res = port.train(lambda e, r=remaining: r[e])
        used = res["train/early_stop_epoch"] + 1 if res["train/early_stop_epoch"] < 3 else 3
        per_iter.append(np.array(remaining[:used])); k += used
    assert k == len(perms), (k, len(perms))
    flat = lambda a: a.reshape(N, T, -1).swapaxes(0, 1).reshape(T, N, -1) if a.shape[0] == N * T else a.reshape(T, N, -1)
    keys = ("observations", "orig_observations", "new_observations", "actions", "rewards", "costs", "orig_costs", "dones",
            "log_probs", "reward_values", "cost_values", "reward_advantages", "cost_advantages", "reward_returns", "cost_returns")
    for key in keys:
        worst[key] = maxdiff(flat(getattr(rb, key)), getattr(b, key).reshape(T, N, -1))
    for key in ("train/nu", "train/average_cost", "train/approx_kl", "train/loss", "train/policy_gradient_loss", "train/entropy_loss"):
        worst[key] = abs(logs[key] - float(res[key]))
    for kname, p in port.policy.params.items():
        worst["w/" + kname] = maxdiff(p.detach().numpy(), agent.policy.state_dict()[kname].numpy())
    print("  port vs reference, max abs deviation:", {k: v for k, v in worst.items() if v > 0} or "all exactly 0")
    assert max(worst.values()) < 1e-5, worst
    pad = lambda a: np.concatenate([a, -np.ones((3 - a.shape[0],) + a.shape[1:], a.dtype)]) if a.shape[0] < 3 else a
    save("g10_lap_grid", **out, learn_actions=acts, perms0=pad(per_iter[0]), perms1=pad(per_iter[1]),
         n_perms=np.array([len(per_iter[0]), len(per_iter[1])]),
         **{f"w0/{k}": v for k, v in sd0.items()}, **{f"cn/{k}": v for k, v in cn0.items()},
         **{f"w1/{k}": v.numpy() for k, v in agent.policy.state_dict().items()},
         **{f"buf/{k}": flat(getattr(rb, k)) for k in keys},
         **{"log/" + k.split("/")[1]: v for k, v in logs.items()})


def fixtures_expert():
    """Re-pack the expert artefacts the runs need (data, not source): HC expert rollouts 0-9 and the expert agent's policy."""
    print("expert fixtures")
    obs, acs = [], []
    for i in range(10):
        d = pickle.load(open(f"{REF}/icrl/expert_data/HCWithPos-New/files/EXPERT/rollouts/{i}.pkl", "rb"))
        obs.append(d["observations"]); acs.append(d["actions"])
    z = zipfile.ZipFile(f"{REF}/icrl/expert_data/HCWithPos-New/files/best_model.zip")
    sd = th.load(io.BytesIO(z.read("policy.pth")))
    save("expert_hc", observations=np.concatenate(obs).astype(np.float32), actions=np.concatenate(acs).astype(np.float32),
         **{f"policy/{k}": v.numpy() for k, v in sd.items()})
    # AntWall (configs[2] / [4]): 5 of the reference's expert rollouts (float32) and the expert agent's policy
    obs, acs, rews, lens = [], [], [], []
    for i in range(5):
        d = pickle.load(open(f"{REF}/icrl/expert_data/AntWall/files/EXPERT/rollouts/{i}.pkl", "rb"))
        obs.append(d["observations"]); acs.append(d["actions"]); rews.append(float(d["rewards"][0])); lens.append(int(d["lengths"][0]))
    z = zipfile.ZipFile(f"{REF}/icrl/expert_data/AntWall/files/best_model.zip")
    sd = th.load(io.BytesIO(z.read("policy.pth")))
    save("expert_ant", observations=np.concatenate(obs).astype(np.float32), actions=np.concatenate(acs).astype(np.float32),
         rewards=np.array(rews), lengths=np.array(lens), **{f"policy/{k}": v.numpy() for k, v in sd.items()})
    # the AntWall -> AntBroken transfer checkpoint (configs[4]; README.md:78) re-packed as arrays for the tests that need the weights
    raw = th.load(f"{REF}/icrl/expert_data/ConstraintTransfer/ICRL/AntBroken/files/best_cn_model.pt")
    save("cn_antbroken", obs_dim=raw["obs_dim"], acs_dim=raw["acs_dim"], is_discrete=raw["is_discrete"],
         hidden_sizes=np.array(raw["hidden_sizes"]), clip_obs=raw["clip_obs"],
         **{f"cn_network/{k}": v.numpy() for k, v in raw["cn_network"].items()})
    # artefacts exactly as the REFERENCE wrote them (bytes of its data files, not source): the on-disk formats the build must read
    import shutil
    art = os.path.join(OUT, "ref_artifacts"); os.makedirs(art, exist_ok=True)
    for src, dst in ((f"{REF}/icrl/expert_data/HCWithPos-New/files/EXPERT/rollouts/0.pkl", "hc_expert_rollout_0.pkl"),
                     (f"{REF}/icrl/expert_data/ConstraintTransfer/ICRL/AntBroken/files/best_cn_model.pt", "antbroken_best_cn_model.pt"),
                     (f"{REF}/icrl/expert_data/HCWithPos-New/files/best_model.zip", "hc_best_model.zip"),
                     (f"{REF}/icrl/expert_data/HCWithPos-New/files/train_env_stats.pkl", "hc_train_env_stats.pkl")):
        shutil.copyfile(src, os.path.join(art, dst)); os.chmod(os.path.join(art, dst), 0o644)
        print(f"  copied {dst} ({os.path.getsize(os.path.join(art, dst)) / 1024:.0f} KB)")
    # what the reference itself reads out of them (expected values for the readers' tests)
    from stable_baselines3.common.vec_env import VecNormalize as RefVecNormalize
    agent = PPOLagrangian.load(os.path.join(art, "hc_best_model.zip"))
    robs = np.random.RandomState(3).randn(7, 18)
    with th.no_grad():
        v_r, v_c, lp, ent = agent.policy.evaluate_actions(th.tensor(robs, dtype=th.float32), th.zeros(7, 6))
    opt = agent.policy.optimizer.state_dict()
    with open(os.path.join(art, "hc_train_env_stats.pkl"), "rb") as f:
        vn = pickle.load(f)
    save("ref_artifacts_expected", probe_obs=robs, v_r=v_r.numpy(), v_c=v_c.numpy(), log_prob=lp.numpy(), entropy=ent.numpy(),
         nu=agent.dual.nu().item(), num_timesteps=agent.num_timesteps, n_updates=agent._n_updates, n_envs=agent.n_envs,
         adam_step=float(opt["state"][0]["step"]), exp_avg_0=opt["state"][0]["exp_avg"].numpy(), exp_avg_sq_5=opt["state"][5]["exp_avg_sq"].numpy(),
         lr=opt["param_groups"][0]["lr"], eps=opt["param_groups"][0]["eps"],
         obs_rms_mean=vn.obs_rms.mean, obs_rms_var=vn.obs_rms.var, obs_rms_count=vn.obs_rms.count, ret_rms_var=vn.ret_rms.var,
         cost_rms_var=getattr(vn, "cost_rms", vn.ret_rms).var, clip_obs=vn.clip_obs, norm_flags=np.array([vn.norm_obs, vn.norm_reward, getattr(vn, "norm_cost", False)]),
         vn_class=np.array(type(vn).__name__))
    # LapGridWorld: the reference ships the expert agent but no rollouts (icrl/expert_data/LGW/files has no EXPERT/rollouts);
    # they are produced the way icrl/run_policy.py does — 20 sampled episodes of that agent — on the restated env.
    z = zipfile.ZipFile(f"{REF}/icrl/expert_data/LGW/files/best_model.zip")
    sd = th.load(io.BytesIO(z.read("policy.pth")))
    stack = o_loop.make_stack(1, "lgw", 0, training=False, norm_obs=False, norm_reward=False, norm_cost=False)
    port = o_loop.PortAgent(stack, seed=0, discrete=True)
    port.policy.load_state_dict(sd)
    th.manual_seed(0)
    oo, o, a, r, l = o_loop.sample_from_agent(port, stack, 20)
    print("  LGW expert: episode rewards", sorted(set(r.tolist())), "backward moves", int(np.sum(a)))
    save("expert_lgw", observations=oo.astype(np.float64), actions=a.astype(np.float32), rewards=r, lengths=l,
         **{f"policy/{k}": v.numpy() for k, v in sd.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g9", "g10", "expert", "g8", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18"]
    table = dict(g1=g1_gae, g2=g2_cost_function, g3=g3_vecnormalize, g4=g4_ppo_minibatch, g5=g5_dual,
                 g6=g6_constraint_net_train, g7=g7_constraint_net_minibatch, g8=g8_icrl_lgw, g11=g11_pid, g12=g12_gail, g13=g13_widths, g14=g14_batch256, g15=g15_wide, g16=g16_batch512, g17=g17_trunk, g18=g18_deep, g9=g9_learn_iteration, g10=g10_lap_grid, expert=fixtures_expert)
    for w in which:
        table[w]()
