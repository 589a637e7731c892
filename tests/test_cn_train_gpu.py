"""GPU parity of ConstraintNet.train (icrl_cn_train) vs the reference (tests/golden/g6) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import cn as o_cn, nets as o_nets

pytestmark = pytest.mark.gpu


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


def _close(a, b, rtol=2e-4, atol=2e-5):
    if np.isnan(b):
        return np.isnan(a)
    if np.isinf(b):
        return a == b
    return abs(a - b) <= atol + rtol * abs(b)


@pytest.mark.parametrize("case", ["psis", "episode", "overflow", "earlystop"])
def test_cn_train_golden(golden, case):
    from icrl_amd.constraint_net import ConstraintNet
    g = _sub(golden("g6_constraint_net"), case + "/")
    psis, iters, tk_on, tk_no, lr = g["cfg"]
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], None, lambda x: float(lr), g["exp_obs"], g["exp_acs"], False, 0.5, clip_obs=20,
                       action_low=lo, action_high=-lo, per_step_importance_sampling=bool(psis),
                       target_kl_old_new=float(tk_on), target_kl_new_old=float(tk_no))
    cn.load_state_dict(_sub(g, "w0/"))
    m = cn.train(int(iters), g["nom_obs"], g["nom_acs"], g["lengths"])
    assert m["backward/early_stop_itr"] == int(g["m/early_stop_itr"])
    for k, v in m.items():
        ref = float(g["m/" + k.split("/")[1]])
        if case == "overflow" and k in ("backward/kl_old_new", "backward/kl_new_old"):
            continue    # float32 episode products overflow: the reference itself yields inf/nan here (order-dependent)
        assert _close(v, ref, rtol=2e-3, atol=2e-4), (k, v, ref)
    for k, v in cn.state_dict().items():
        assert np.allclose(v.numpy(), g["w1/" + k], rtol=2e-3, atol=2e-4), (k, np.abs(v.numpy() - g["w1/" + k]).max())


@pytest.mark.parametrize("kind,hidden,psis,Nn,Ne,eplen", [("hc", [20], True, 3000, 1500, 1000), ("ant", [40, 40], True, 1500, 900, 500),
                                                          ("hc", [20], False, 400, 300, 100),
                                                          # layers above 64 units (-cl 128 128 / -cl 100: create_mlp takes any width,
                                                          # torch_layers.py:93-126): weights read from device memory, 64 rows per workgroup
                                                          ("hc", [128, 128], True, 3000, 1500, 1000), ("ant", [128, 96], True, 1500, 900, 500),
                                                          ("hc", [100], False, 400, 300, 100),
                                                          # more than two hidden layers (-cl 64 64 64 ...): one activation image per layer
                                                          # and two alternating gradient images in the LDS
                                                          ("hc", [64, 64, 64], True, 3000, 1500, 1000), ("ant", [64, 48, 64, 32], True, 1500, 900, 500),
                                                          ("hc", [96, 96, 96], False, 400, 300, 100), ("hc", [24, 20, 16, 12], True, 1000, 640, 500),
                                                          ("hc", [], True, 1000, 640, 500)])      # `-cl` without widths: sigmoid(Linear)
def test_cn_train_vs_oracle(kind, hidden, psis, Nn, Ne, eplen):
    from icrl_amd.constraint_net import ConstraintNet
    rng = np.random.RandomState(Nn)
    od, ad = (18, 6) if kind == "hc" else (113, 8)
    lo = -np.ones(ad, np.float32)
    exp_obs, exp_acs = rng.randn(Ne, od), rng.uniform(-1, 1, (Ne, ad)).astype(np.float32)
    nom_obs, nom_acs = rng.randn(Nn, od) * 1.5, rng.uniform(-1.2, 1.2, (Nn, ad))
    lengths = np.array([eplen] * (Nn // eplen))
    torch.manual_seed(1)
    orc = o_nets.CostNet(od, ad, hidden, False, None, None, 20, lo, -lo)
    cn = ConstraintNet(od, ad, hidden, None, lambda x: 0.01, exp_obs, exp_acs, False, 0.6, clip_obs=20, action_low=lo, action_high=-lo,
                       per_step_importance_sampling=psis, target_kl_old_new=10, target_kl_new_old=2.5)
    cn.load_state_dict(orc.state_dict())
    opt = torch.optim.Adam(orc.parameters(), lr=0.01, eps=1e-5)
    ref_cost = orc.cost_function(nom_obs[:777], nom_acs[:777])      # on the initial weights (wide nets: cn_cost_rows_kernel)
    om = o_cn.cn_train(orc, opt, 5, orc.prepare(nom_obs, nom_acs), orc.prepare(exp_obs, exp_acs), lengths, reg_coeff=0.6,
                       per_step=psis, target_kl_old_new=10, target_kl_new_old=2.5, factored=True)
    assert np.allclose(cn.cost_function(nom_obs[:777], nom_acs[:777].astype(np.float32)), ref_cost, rtol=2e-5, atol=2e-6)
    m = cn.train(5, nom_obs, nom_acs, lengths)
    assert np.array_equal(cn.prepare_data(nom_obs, nom_acs).cpu().numpy(), orc.prepare(nom_obs, nom_acs).numpy())
    assert m["backward/early_stop_itr"] == om["backward/early_stop_itr"]
    for k, v in m.items():
        assert _close(v, float(om[k]), rtol=3e-3, atol=3e-4), (k, v, om[k])
    for k, v in cn.state_dict().items():
        ref = orc.params[k].detach().numpy()
        assert np.allclose(v.numpy(), ref, rtol=3e-3, atol=3e-4), (k, np.abs(v.numpy() - ref).max())
    # a second call continues the optimiser state (Adam step counter, moments)
    assert cn.adam_step == (5 if m["backward/early_stop_itr"] == 5 else m["backward/early_stop_itr"])


def test_policy_evaluate_actions(golden):
    from icrl_amd.policies import ActorTwoCriticsPolicy
    from icrl_amd import spaces
    g = golden("expert_hc")
    pol = ActorTwoCriticsPolicy(spaces.Box(-np.inf, np.inf, (18,), np.float64), spaces.Box(-1, 1, (6,), np.float32))
    pol.load_state_dict(_sub(g, "policy/"))
    op = o_nets.TwoCriticPolicy(18, 6); op.load_state_dict(_sub(g, "policy/"))
    obs, act = g["observations"][:257], g["actions"][:257]
    vr, vc, lp, ent = pol.evaluate_actions(obs, act)
    with torch.no_grad():
        ovr, ovc, olp, oent = op.evaluate_actions(torch.as_tensor(obs), torch.as_tensor(act))
    assert np.allclose(lp.cpu().numpy(), olp.numpy(), rtol=2e-4, atol=2e-4)
    assert np.allclose(vr.cpu().numpy(), ovr.numpy(), rtol=2e-4, atol=2e-4)
    assert np.allclose(ent.cpu().numpy(), oent.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind,n", [("hc", 257), ("hc", 1000), ("ant", 333), ("narrow", 100)])
def test_policy_rows_kernel_equals_one_workgroup_per_row(kind, n):
    """icrl_policy_forward / icrl_policy_evaluate over >= 64 rows run 16 rows per workgroup pass as fp32 MFMA tiles
    (policy_rows_kernel); below, one workgroup per row with per-lane fmaf chains.  Same values bit for bit: evaluate / forward
    the rows at once, and in chunks of 50."""
    from icrl_amd.policies import ActorTwoCriticsPolicy
    from icrl_amd import spaces
    od, ad = (113, 8) if kind == "ant" else (18, 6)
    torch.manual_seed(3)
    kw = dict(net_arch=[dict(pi=[40, 24], vf=[64, 20], cvf=[16, 64])]) if kind == "narrow" else {}
    pol = ActorTwoCriticsPolicy(spaces.Box(-np.inf, np.inf, (od,), np.float64), spaces.Box(-1, 1, (ad,), np.float32), **kw)
    rng = np.random.RandomState(5)
    obs = rng.randn(n, od) * 2.0
    act = rng.randn(n, ad).astype(np.float32)
    noise = rng.randn(n, ad).astype(np.float32)
    whole = [t.cpu().numpy() for t in pol.evaluate_actions(obs, act)]
    parts = [pol.evaluate_actions(obs[i:i + 50], act[i:i + 50]) for i in range(0, n, 50)]
    for k, name in enumerate(("reward_values", "cost_values", "log_prob", "entropy")):
        ref = np.concatenate([pt[k].cpu().numpy() for pt in parts])
        assert np.array_equal(whole[k], ref), (name, np.abs(whole[k] - ref).max())
    for det in (False, True):
        a, vr, vc, lp = pol.forward(obs, deterministic=det, noise=noise)
        cl = pol.last_clipped
        got = [t.cpu().numpy() for t in (a, vr, vc, lp, cl)]
        ref = [[], [], [], [], []]
        for i in range(0, n, 50):
            a2, vr2, vc2, lp2 = pol.forward(obs[i:i + 50], deterministic=det, noise=noise[i:i + 50])
            for lst, t in zip(ref, (a2, vr2, vc2, lp2, pol.last_clipped)):
                lst.append(t.cpu().numpy())
        for g_, r_, name in zip(got, ref, ("actions", "reward_values", "cost_values", "log_prob", "clipped")):
            assert np.array_equal(g_, np.concatenate(r_)), (det, name)


@pytest.mark.parametrize("case", ["mb_psis", "mb_episode", "mb_nois", "mb_gail"])
def test_cn_train_minibatch_golden(golden, case):
    """`--cn_batch_size` mode against the REFERENCE's train() with the recorded permutations (tests/golden/g7)."""
    from icrl_amd.constraint_net import ConstraintNet
    g = _sub(golden("g7_constraint_net_minibatch"), case + "/")
    psis, iters, tk_on, tk_no, lr, bs, nois, gail = g["cfg"]
    lo = -np.ones(6, np.float32)
    cn = ConstraintNet(18, 6, [20], int(bs), lambda x: float(lr), g["exp_obs"], g["exp_acs"], False, 0.5, clip_obs=20,
                       action_low=lo, action_high=-lo, per_step_importance_sampling=bool(psis), no_importance_sampling=bool(nois),
                       train_gail_lambda=bool(gail), target_kl_old_new=float(tk_on), target_kl_new_old=float(tk_no))
    cn.load_state_dict(_sub(g, "w0/"))
    perms = g["perms"]
    full = np.zeros((int(iters), perms.shape[1]), np.int64)
    full[:len(perms)] = perms
    m = cn.train(int(iters), g["nom_obs"], g["nom_acs"], g["lengths"], perms=full)
    n_batches = -(-perms.shape[1] // int(bs))
    assert cn.adam_step == len(perms) * n_batches
    for k, v in m.items():
        ref = float(g["m/" + k.split("/")[1]])
        assert _close(v, ref, rtol=2e-3, atol=2e-4), (k, v, ref)
    for k, v in cn.state_dict().items():
        assert np.allclose(v.numpy(), g["w1/" + k], rtol=2e-3, atol=3e-4), (k, np.abs(v.numpy() - g["w1/" + k]).max())


def test_cn_train_minibatch_default_stream_matches_oracle():
    """without `perms` the batches come from np.random.permutation exactly like the reference's get()."""
    from icrl_amd.constraint_net import ConstraintNet
    rng = np.random.RandomState(9)
    lo = -np.ones(6, np.float32)
    exp_obs, exp_acs = rng.randn(200, 18), rng.uniform(-1, 1, (200, 6)).astype(np.float32)
    nom_obs, nom_acs = rng.randn(300, 18), rng.uniform(-1.2, 1.2, (300, 6))
    lengths = np.array([100, 100, 100])
    torch.manual_seed(2)
    orc = o_nets.CostNet(18, 6, [20], False, None, None, 20, lo, -lo)
    cn = ConstraintNet(18, 6, [20], 64, lambda x: 0.01, exp_obs, exp_acs, False, 0.5, clip_obs=20, action_low=lo, action_high=-lo,
                       per_step_importance_sampling=True, target_kl_old_new=10, target_kl_new_old=2.5)
    cn.load_state_dict(orc.state_dict())
    opt = torch.optim.Adam(orc.parameters(), lr=0.01, eps=1e-5)
    np.random.seed(123)
    om = o_cn.cn_train(orc, opt, 4, orc.prepare(nom_obs, nom_acs), orc.prepare(exp_obs, exp_acs), lengths, reg_coeff=0.5,
                       per_step=True, target_kl_old_new=10, target_kl_new_old=2.5, batch_size=64)
    after_oracle = np.random.rand()
    np.random.seed(123)
    m = cn.train(4, nom_obs, nom_acs, lengths)
    assert np.random.rand() == after_oracle            # the global stream is left where the reference leaves it
    for k, v in m.items():
        assert _close(v, float(om[k]), rtol=3e-3, atol=3e-4), (k, v, om[k])
    for k, v in cn.state_dict().items():
        ref = orc.params[k].detach().numpy()
        assert np.allclose(v.numpy(), ref, rtol=3e-3, atol=3e-4), (k, np.abs(v.numpy() - ref).max())


from helpers.arches import ARCHES, oracle_arch_kwargs


@pytest.mark.parametrize("kind,n", [("hc", 64), ("hc", 1000), ("ant", 333), ("narrow", 200), ("wide", 40), ("wide", 300),
                                    ("trunk", 150), ("deep", 70), ("trunk-only", 33), ("bare", 20)])
def test_policy_rows_kernel_vs_oracle(kind, n):
    """policy_rows_kernel (>= 64 rows: 16 rows per workgroup pass as fp32 MFMA tiles; the KL metrics' evaluate_actions and batched
    predict) against the oracle's ActorTwoCriticsPolicy.evaluate_actions / forward directly (policies.py:716-731, 752-767;
    distributions.py:143-171) — not only against the one-workgroup-per-row kernel."""
    from icrl_amd.policies import ActorTwoCriticsPolicy
    from icrl_amd import spaces
    from oracle import nets as o_nets
    od, ad = (113, 8) if kind == "ant" else (18, 6)
    torch.manual_seed(11)
    arch = dict(pi=[40, 24], vf=[64, 20], cvf=[16, 64]) if kind == "narrow" else None
    if kind == "wide":      # layers above 64: policy_generic_kernel (csrc/generic.hip)
        arch = dict(pi=[128, 100], vf=[72, 128], cvf=[200, 256])
    net_arch = ARCHES.get(kind, [arch] if arch else None)      # (ARCHES: the table-driven generic kernel with an `arch` descriptor)
    kw = dict(net_arch=net_arch) if net_arch else {}
    pol = ActorTwoCriticsPolicy(spaces.Box(-np.inf, np.inf, (od,), np.float64), spaces.Box(-1, 1, (ad,), np.float32), **kw)
    assert pol.wide == (kind == "wide" or kind in ARCHES) and (pol.kind == "arch") == (kind in ARCHES)
    op = o_nets.TwoCriticPolicy(od, ad, **(oracle_arch_kwargs(net_arch) if net_arch else {}))
    assert list(op.params) == list(pol.shapes) and all(tuple(op.params[k].shape) == pol.shapes[k] or pol.kind != "arch" for k in op.params)
    op.load_state_dict(pol.state_dict())
    rng = np.random.RandomState(9)
    obs = (rng.randn(n, od) * 2.0).astype(np.float32)
    act = rng.randn(n, ad).astype(np.float32)
    noise = rng.randn(n, ad).astype(np.float32)
    vr, vc, lp, ent = [t.cpu().numpy().reshape(-1) for t in pol.evaluate_actions(obs, act)]
    with torch.no_grad():
        o_vr, o_vc, o_lp, o_ent = op.evaluate_actions(torch.as_tensor(obs), torch.as_tensor(act))
    for name, got, ref in (("reward_values", vr, o_vr), ("cost_values", vc, o_vc), ("log_prob", lp, o_lp), ("entropy", ent, o_ent)):
        ref = ref.numpy().reshape(-1)
        assert np.allclose(got, ref, rtol=1e-5, atol=2e-6), (name, np.abs(got - ref).max())
    a, fvr, fvc, flp = [t.cpu().numpy() for t in pol.forward(obs, deterministic=False, noise=noise)]
    with torch.no_grad():
        o_a, o_fvr, o_fvc, o_flp = op.forward(torch.as_tensor(obs), noise=torch.as_tensor(noise))
    for name, got, ref in (("actions", a, o_a), ("reward_values", fvr, o_fvr), ("cost_values", fvc, o_fvc), ("log_prob", flp, o_flp)):
        ref = ref.numpy().reshape(got.shape)
        assert np.allclose(got, ref, rtol=1e-5, atol=2e-6), (name, np.abs(got - ref).max())
