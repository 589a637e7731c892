"""GPU: the fine-grained entry points of the update (csrc/fine.hip) driven the way a maintainer of the
reference would bind them — the MLPs and autograd stay in torch (here: the oracle's torch policy on the CPU standing in for
`policy.evaluate_actions`), the library computes the loss terms and d loss / d outputs, then clip_grad_norm_ + Adam on the flat
parameter buffer — against the reference's own numbers (tests/golden/g4: its policy / optimizer objects stepped 3 x on one batch)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nets as o_nets, ppo as o_ppo

pytestmark = pytest.mark.gpu


def _sub(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def _dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda().contiguous()


def test_loss_fwd_bwd_and_clip_adam_reproduce_the_reference_steps(golden):
    """ppo_lag.py:216-288 with the network in torch and everything else in the library: three consecutive optimiser steps."""
    from icrl_amd import _lib, structs as S
    L = _lib.lib()
    g = golden("g4_ppo_minibatch")
    B = int(g["obs"].shape[0])
    pol = o_nets.TwoCriticPolicy(18, 6)
    pol.load_state_dict(_sub(g, "w0/"))
    names = list(pol.params)
    sizes = [int(pol.params[k].numel()) for k in names]
    n = sum(sizes)
    flat = _dev(np.concatenate([pol.params[k].detach().numpy().ravel() for k in names]))
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    work, out2, terms = torch.zeros(256, device="cuda"), torch.zeros(2, device="cuda"), torch.zeros(8, device="cuda")
    hp = S.PpoHyperT(B, 1, 0, 0)
    hp.clip_range, hp.ent_coef, hp.reward_vf_coef, hp.cost_vf_coef, hp.max_grad_norm = float(g["clip"]), 0.0, 0.5, 0.5, 0.5
    hp.clip_range_reward_vf = hp.clip_range_cost_vf = -1.0
    hp.lr, hp.adam_beta1, hp.adam_beta2, hp.adam_eps = float(g["lr"]), 0.9, 0.999, 1e-5
    nu = _dev(np.float32([float(g["nu"])]))
    dev = {k: _dev(g[k]) for k in ("old_lp", "adv_r", "adv_c", "ret_r", "ret_c")}
    obs, act = torch.as_tensor(g["obs"]), torch.as_tensor(g["act"])
    st = _lib.current_stream()
    for s in range(3):
        # ---- the host's part: forward through ITS networks (torch, autograd on)
        for k in names:
            pol.params[k].grad = None
        v_r, v_c, lp, ent = pol.evaluate_actions(obs, act)
        v_r, v_c = v_r.flatten(), v_c.flatten()
        ent_t = ent if ent is not None and ent.dim() > 0 else None
        # ---- the library's part 1: loss terms + d loss / d outputs
        d_lp, d_vr, d_vc, d_en = (torch.empty(B, device="cuda") for _ in range(4))
        o_lp, o_vr, o_vc = _dev(lp.detach().numpy()), _dev(v_r.detach().numpy()), _dev(v_c.detach().numpy())      # (kept alive across the call)
        o_en = _dev(ent_t.detach().numpy()) if ent_t is not None else None
        args = [_lib.ptr(o_lp), _lib.ptr(dev["old_lp"]), _lib.ptr(dev["adv_r"]), _lib.ptr(dev["adv_c"]),
                _lib.ptr(o_vr), _lib.ptr(o_vc), _lib.ptr(dev["ret_r"]), _lib.ptr(dev["ret_c"]), None, None,
                _lib.ptr(o_en), _lib.ptr(nu), ctypes.byref(hp), B, _lib.ptr(terms),
                _lib.ptr(d_lp), _lib.ptr(d_vr), _lib.ptr(d_vc), _lib.ptr(d_en) if ent_t is not None else None, st]
        _lib.check(L.icrl_ppo_lag_loss_fwd_bwd(*args), "icrl_ppo_lag_loss_fwd_bwd")
        t = terms.cpu().numpy()
        for i, key in enumerate(("loss", "policy_loss", "rvl", "cvl", "entropy_loss", "approx_kl", "clip_fraction")):
            ref = float(g[f"s{s}/{key}"])
            assert abs(t[i] - ref) <= 2e-6 + 2e-5 * abs(ref), (s, key, t[i], ref)
        # ---- the host's part: backward through its networks from the library's output gradients
        outs, grads = [lp, v_r, v_c], [d_lp.cpu(), d_vr.cpu(), d_vc.cpu()]
        if ent_t is not None:
            outs.append(ent_t); grads.append(d_en.cpu())
        torch.autograd.backward(outs, grads)
        for k in names:
            ref = g[f"s{s}/grad/{k}"]
            got = pol.params[k].grad.numpy()
            assert np.allclose(got, ref, rtol=2e-4, atol=2e-7), (s, k, np.abs(got - ref).max())
        # ---- the library's part 2: clip_grad_norm_ + Adam on the flat buffer
        gflat = _dev(np.concatenate([pol.params[k].grad.numpy().ravel() for k in names]))
        _lib.check(L.icrl_clip_adam_step(_lib.ptr(flat), _lib.ptr(gflat), _lib.ptr(m), _lib.ptr(v), _lib.ptr(step), n, ctypes.byref(hp), _lib.ptr(work),
                                         _lib.ptr(out2), st), "icrl_clip_adam_step")
        assert abs(out2[0].item() - float(g[f"s{s}/grad_norm"])) <= 1e-5 * max(1.0, float(g[f"s{s}/grad_norm"]))
        new = flat.cpu().numpy()
        off = 0
        with torch.no_grad():
            for k, sz in zip(names, sizes):
                got = new[off:off + sz].reshape(pol.params[k].shape)
                ref = g[f"s{s}/after/{k}"]
                assert np.allclose(got, ref, rtol=1e-5, atol=3e-7), (s, k, np.abs(got - ref).max())
                pol.params[k].copy_(torch.as_tensor(got))
                off += sz
    assert int(step.item()) == 3


def test_minibatch_gather_and_adv_stats():
    """buffers.py:53-65,594-627: flat env-major indices -> rows of the [T, N] buffer; ppo_lag.py:219-222 statistics."""
    from icrl_amd import _lib, spaces
    from icrl_amd.buffers import RolloutBufferWithCost
    L = _lib.lib()
    T, N, od, ad, n = 40, 7, 18, 6, 100
    rb = RolloutBufferWithCost(T, spaces.Box(-np.inf, np.inf, (od,), np.float64), spaces.Box(-1, 1, (ad,), np.float32), "cuda", n_envs=N)
    rng = np.random.RandomState(1)
    ref = {}
    for k in ("observations", "actions", "log_probs", "reward_advantages", "cost_advantages", "reward_returns", "cost_returns", "reward_values", "cost_values"):
        a = rng.randn(*getattr(rb, k).shape).astype(np.float32)
        getattr(rb, k).copy_(torch.as_tensor(a))
        ref[k] = o_ppo.env_major(a.reshape(T, N, -1))
    idx = rng.permutation(T * N)[:n].astype(np.int32)
    outs = dict(obs=torch.empty(n, od, device="cuda"), act=torch.empty(n, ad, device="cuda"))
    for k in ("lp", "ar", "ac", "rr", "rc", "vr", "vc"):
        outs[k] = torch.empty(n, device="cuda")
    s = rb.struct()
    d_idx = _dev(idx, torch.int32)
    _lib.check(L.icrl_minibatch_gather(ctypes.byref(s), _lib.ptr(d_idx), n, *[_lib.ptr(outs[k]) for k in ("obs", "act", "lp", "ar", "ac", "rr", "rc", "vr", "vc")],
                                       _lib.current_stream()), "icrl_minibatch_gather")
    pairs = dict(obs="observations", act="actions", lp="log_probs", ar="reward_advantages", ac="cost_advantages", rr="reward_returns", rc="cost_returns",
                 vr="reward_values", vc="cost_values")
    for k, name in pairs.items():
        assert np.array_equal(outs[k].cpu().numpy().reshape(n, -1), ref[name][idx].reshape(n, -1)), k
    out4 = torch.zeros(4, device="cuda")
    _lib.check(L.icrl_adv_stats(_lib.ptr(outs["ar"]), _lib.ptr(outs["ac"]), n, _lib.ptr(out4), _lib.current_stream()), "icrl_adv_stats")
    ar, ac = torch.as_tensor(ref["reward_advantages"][idx].ravel()), torch.as_tensor(ref["cost_advantages"][idx].ravel())
    exp = [ar.mean().item(), 1.0 / (ar.std().item() + 1e-8), ac.mean().item(), ar.std().item()]
    assert np.allclose(out4.cpu().numpy(), exp, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("case", "abcd")
def test_dual_step_follows_the_reference_trajectories(golden, case):
    """dual_variable.py:9-57 on the device against the reference's own nu trajectories (tests/golden/g5: 200 costs each, incl. the
    clamp-floor case); float32 arithmetic with libm's expf / log1pf instead of torch's: 2e-6 relative."""
    from icrl_amd import _lib
    from icrl_amd.dual_variable import DualVariable, _inv_softplus_floor
    L = _lib.lib()
    g = _sub(golden("g5_dual"), case + "/")
    nu0, lr, budget = (float(x) for x in g["params"])
    host = DualVariable(budget, lr, nu0, None)
    state = _dev(np.float32([host.log_nu, 0, 0, host.nu().item()]))
    t = torch.zeros(1, dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda")
    clamp = float(np.float32(_inv_softplus_floor(host.clamp_at)))
    traj = []
    for c in g["costs"]:
        _lib.check(L.icrl_dual_step(_lib.ptr(state), _lib.ptr(t), None, float(c), budget, lr, clamp, _lib.ptr(loss), _lib.current_stream()), "icrl_dual_step")
        traj.append(torch.cat([state[3:4], loss, state[0:1]]).clone())
    got = torch.stack(traj).cpu().numpy()
    assert int(t.item()) == len(g["costs"])
    assert np.allclose(got[:, 0], g["traj"][:, 0], rtol=2e-6, atol=1e-7), np.abs(got[:, 0] - g["traj"][:, 0]).max()      # nu
    assert np.allclose(got[:, 2], g["traj"][:, 2], rtol=2e-6, atol=2e-7)                                                   # log_nu
    assert np.allclose(got[:, 1], g["traj"][:, 1], rtol=2e-5, atol=1e-7)                                                   # loss = -nu (cost - budget)


def test_buffer_add_writes_row_t():
    """buffers.py:554-592: one step's arrays -> row t of every plane (float64 observations / rewards stored as float32)."""
    from icrl_amd import _lib, spaces
    from icrl_amd.buffers import RolloutBufferWithCost
    L = _lib.lib()
    T, N, od, ad, t = 6, 5, 18, 6, 3
    rb = RolloutBufferWithCost(T, spaces.Box(-np.inf, np.inf, (od,), np.float64), spaces.Box(-1, 1, (ad,), np.float32), "cuda", n_envs=N)
    rng = np.random.RandomState(2)
    host = dict(obs=rng.randn(N, od), orig_obs=rng.randn(N, od), new_obs=rng.randn(N, od), new_orig_obs=rng.randn(N, od), action=rng.randn(N, ad).astype(np.float32),
                reward=rng.randn(N), cost=rng.rand(N), orig_cost=rng.rand(N).astype(np.float32), done=(rng.rand(N) < 0.5).astype(np.uint8),
                reward_value=rng.randn(N).astype(np.float32), cost_value=rng.randn(N).astype(np.float32), log_prob=rng.randn(N).astype(np.float32))
    dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in host.items()}
    s = rb.struct()
    order = ("obs", "orig_obs", "new_obs", "new_orig_obs", "action", "reward", "cost", "orig_cost", "done", "reward_value", "cost_value", "log_prob")
    _lib.check(L.icrl_buffer_add(ctypes.byref(s), t, *[_lib.ptr(dev[k]) for k in order], _lib.current_stream()), "icrl_buffer_add")
    planes = dict(obs="observations", orig_obs="orig_observations", new_obs="new_observations", new_orig_obs="new_orig_observations", action="actions",
                  reward="rewards", cost="costs", orig_cost="orig_costs", done="dones", reward_value="reward_values", cost_value="cost_values", log_prob="log_probs")
    for k, name in planes.items():
        got = getattr(rb, name).cpu().numpy()
        assert np.array_equal(got[t].reshape(N, -1), host[k].astype(np.float32).reshape(N, -1)), k
        assert not got[:t].any() and not got[t + 1:].any(), k          # only row t was touched
    assert L.icrl_buffer_add(ctypes.byref(s), T, *[_lib.ptr(dev[k]) for k in order], _lib.current_stream()) == 1       # refused: t outside the buffer
    L.icrl_clear_error()


@pytest.mark.parametrize("per_step", [False, True])
def test_is_weights_vs_oracle(per_step):
    """constraint_net.py:231-256 against the oracle restatement (pinned to the reference by g6): ragged episodes."""
    from icrl_amd import _lib
    from oracle import cn as o_cn
    L = _lib.lib()
    rng = np.random.RandomState(3)
    lengths = [7, 40, 1, 23, 64, 12, 33]
    N = sum(lengths)
    old = torch.as_tensor(rng.uniform(0.2, 0.9, (N, 1)).astype(np.float32))
    new = (old + torch.as_tensor(rng.uniform(-0.02, 0.02, (N, 1)).astype(np.float32))).clamp(0.01, 0.99)
    w_ref, kl_on, kl_no = o_cn.is_weights_and_kls(old.clone(), new.clone(), lengths, 1e-5, per_step)
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)).cuda()
    d_old, d_new = old.flatten().cuda().contiguous(), new.flatten().cuda().contiguous()
    w, prod, out4 = torch.zeros(N, device="cuda"), torch.zeros(len(lengths), device="cuda"), torch.zeros(4, device="cuda")
    _lib.check(L.icrl_is_weights(_lib.ptr(d_old), _lib.ptr(d_new), N, _lib.ptr(offs), len(lengths), 1e-5, int(per_step), _lib.ptr(w), _lib.ptr(prod), _lib.ptr(out4),
                                 _lib.current_stream()), "icrl_is_weights")
    assert np.allclose(w.cpu().numpy(), w_ref.numpy().ravel(), rtol=2e-5, atol=1e-7)
    o = out4.cpu().numpy()
    assert abs(o[0] - kl_on.item()) <= 2e-5 * max(1.0, abs(kl_on.item())) and abs(o[1] - kl_no.item()) <= 2e-5 * max(1.0, abs(kl_no.item())) + 1e-6


@pytest.mark.parametrize("mode,weights", [(0, True), (0, False), (2, True), (1, False)])
def test_cn_loss_fwd_bwd_vs_autograd(mode, weights):
    """constraint_net.py:188-202 (and gail_utils.py's BCE) on the network's outputs: loss terms and d loss / d prediction against torch
    autograd of the oracle's expressions.  mode 2: the per-step broadcast quirk (nominal_loss = mean(w) * mean(log zeta))."""
    from icrl_amd import _lib
    from oracle import cn as o_cn
    L = _lib.lib()
    rng = np.random.RandomState(4)
    Bn, Be, reg, eps = 300, 170, 0.5, 1e-5
    nom = torch.as_tensor(rng.uniform(0.05, 0.95, (Bn, 1)).astype(np.float32)).requires_grad_(True)
    exp = torch.as_tensor(rng.uniform(0.05, 0.95, (Be, 1)).astype(np.float32)).requires_grad_(True)
    w = torch.as_tensor(rng.uniform(0.5, 1.5, Bn).astype(np.float32)) if weights else torch.ones(Bn)

    class _Net:                       # the oracle's cn_loss evaluates net.forward(batch): hand it the leaf predictions
        def forward(self, x):
            return x
    is_batch = w[..., None, None] if mode == 2 else w[..., None]
    loss, e_l, n_l, r_l, _, _ = o_cn.cn_loss(_Net(), nom, exp, is_batch, reg, eps, gail=bool(mode & 1), factored=(mode == 2))
    loss.backward()
    d_nom, d_exp, terms = torch.zeros(Bn, device="cuda"), torch.zeros(Be, device="cuda"), torch.zeros(6, device="cuda")
    dn, de = nom.detach().flatten().cuda().contiguous(), exp.detach().flatten().cuda().contiguous()
    dw = w.cuda().contiguous() if weights else None
    _lib.check(L.icrl_cn_loss_fwd_bwd(_lib.ptr(dn), _lib.ptr(de), _lib.ptr(dw), Bn, Be, reg, eps, mode, _lib.ptr(terms), _lib.ptr(d_nom), _lib.ptr(d_exp),
                                      _lib.current_stream()), "icrl_cn_loss_fwd_bwd")
    t = terms.cpu().numpy()
    for got, ref in zip(t[:4], (loss.item(), e_l.item(), n_l.item(), float(r_l))):
        assert abs(got - ref) <= 2e-6 + 2e-5 * abs(ref), (t, loss.item(), e_l.item(), n_l.item(), float(r_l))
    assert np.allclose(d_nom.cpu().numpy(), nom.grad.numpy().ravel(), rtol=2e-5, atol=1e-8)
    assert np.allclose(d_exp.cpu().numpy(), exp.grad.numpy().ravel(), rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("n", [1, 7, 2048 * 64, 2048 * 512 + 3])
def test_explained_variance_vs_the_reference_formula(n):
    """icrl_explained_variance against common/utils.py:43-59 (oracle.loop.explained_variance) in the reference's argument order
    (ppo_lag.py:311-312: y_pred = returns, y_true = values), one and two pairs, and the NaN of a constant y_true."""
    from icrl_amd import _lib
    from oracle.loop import explained_variance
    L = _lib.lib()
    rng = np.random.RandomState(n)
    ret_r, val_r = (3 + 2 * rng.randn(n)).astype(np.float32), (3 + 2 * rng.randn(n)).astype(np.float32)
    val_c = (0.3 * rng.randn(n) + 1).astype(np.float32)
    ret_c = (val_c + 0.1 * rng.randn(n)).astype(np.float32)
    d = [_dev(x) for x in (ret_r, val_r, ret_c, val_c)]
    work, out = torch.zeros(8 * 256, dtype=torch.float64, device="cuda"), torch.full((2,), 7.0, device="cuda")
    _lib.check(L.icrl_explained_variance(_lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), _lib.ptr(d[3]), n, _lib.ptr(work), _lib.ptr(out),
                                         _lib.current_stream()), "icrl_explained_variance")
    got = out.cpu().numpy()
    for g_, (yp, yt) in zip(got, ((ret_r, val_r), (ret_c, val_c))):
        want = explained_variance(yp.astype(np.float64), yt.astype(np.float64))
        if n == 1:
            assert np.isnan(g_) and np.isnan(want)
        else:
            assert abs(g_ - want) <= 1e-6 * max(1.0, abs(want)), (g_, want)
            assert abs(g_ - explained_variance(yp, yt)) <= 1e-4 * max(1.0, abs(want))      # numpy's own float32 evaluation
    # one pair: out[1] untouched; a constant y_true: NaN
    out.fill_(7.0)
    const = torch.full((max(n, 2),), 1.5, device="cuda")
    _lib.check(L.icrl_explained_variance(_lib.ptr(d[0]) if n > 1 else _lib.ptr(const), _lib.ptr(const), None, None, max(n, 2) if n == 1 else n,
                                         _lib.ptr(work), _lib.ptr(out), _lib.current_stream()), "icrl_explained_variance") if n <= 7 else None
    if n <= 7:
        got = out.cpu().numpy()
        assert np.isnan(got[0]) and got[1] == 7.0
