"""GPU parity: icrl_gae_dual (HIP, through the C ABI) vs the oracle and vs the reference's golden vectors."""
import numpy as np
import pytest
import torch

from oracle import gae as o_gae

pytestmark = pytest.mark.gpu


def _run(arrs, params, W=0):
    from icrl_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    f = lambda k: torch.as_tensor(np.ascontiguousarray(arrs[k], dtype=np.float32), device=dev)
    r, c, vr, vc, d = f("rewards"), f("costs"), f("reward_values"), f("cost_values"), f("dones")
    lvr, lvc = f("last_v_r"), f("last_v_c")
    ld = torch.as_tensor(np.asarray(arrs["last_dones"]).astype(np.uint8), device=dev)
    T, N = r.shape
    outs = [torch.full((T, N), float("nan"), device=dev) for _ in range(4)]
    err = L.icrl_gae_dual_ex(*(_lib.ptr(x) for x in (r, c, vr, vc, d, lvr, lvc, ld, *outs)), T, N,
                             *[float(p) for p in params], W, _lib.current_stream())
    _lib.check(err, "icrl_gae_dual_ex")
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs]


def _ulp_diff(a, b):
    ai, bi = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi).max()


_G1_CASES = {"t8n3": 3, "t2000n1": 1, "t256n16": 16, "t1n4": 4, "t64n5": 5}          # case -> N
# launch shapes: library heuristic, 1 / 4 / 16 waves per tile, streaming one column per lane (106), four columns per lane (108, 111: N % 4 == 0)
_G1_PARAMS = [(c, W) for c in _G1_CASES for W in (0, 1, 4, 16, 106, 108, 111) if W <= 106 or _G1_CASES[c] % 4 == 0]


@pytest.mark.parametrize("case,W", _G1_PARAMS)
def test_gae_golden(golden, case, W):
    g = golden("g1_gae")
    arrs = {k.split("/")[1]: g[k] for k in g.files if k.startswith(case + "/")}
    adv_r, adv_c, ret_r, ret_c = _run(arrs, arrs["params"], W)
    for got, key in ((adv_r, "reward_advantages"), (adv_c, "cost_advantages"), (ret_r, "reward_returns"), (ret_c, "cost_returns")):
        if W == 1 or W > 100:
            assert np.array_equal(got, arrs[key]), key          # sequential scan: bit-exact with the reference
        else:
            # time-chunked scan re-associates float64 products: <= 1 ulp of float32 (rtol 1.2e-7), in practice exact
            assert np.allclose(got, arrs[key], rtol=2e-7, atol=1e-7), key


@pytest.mark.parametrize("T,N", [(2048, 64), (2048, 256), (500, 1000), (37, 4097), (2048, 4096), (33, 65536), (9, 131076), (5, 131074), (256, 131072)])
def test_gae_vs_oracle_random(T, N):
    """(256, 131 072): the launch shape bench.py's roofline is quoted on (gae_dual_x4_kernel<4,1>, 512 one-wave workgroups, 64 batches of
    the double-buffered 4-row loop per column).  incl. the streaming shapes the heuristic picks: one column per lane from 65 536 envs on, four columns per lane from 131 072
    on when N % 4 == 0 (131 076: a ragged last wave; 131 074: falls back to one column per lane); all bit-exact."""
    rng = np.random.RandomState(T + N)
    arrs = dict(rewards=rng.randn(T, N), costs=rng.rand(T, N), reward_values=rng.randn(T, N), cost_values=rng.randn(T, N),
                dones=(rng.rand(T, N) < 0.002), last_v_r=rng.randn(N), last_v_c=rng.randn(N), last_dones=rng.rand(N) < 0.2)
    arrs = {k: (v.astype(np.float32) if v.dtype != bool else v) for k, v in arrs.items()}
    params = (0.99, 0.95, 0.99, 0.9)
    o = o_gae.dual_gae(arrs["rewards"], arrs["costs"], arrs["reward_values"], arrs["cost_values"],
                       arrs["dones"].astype(np.float32), arrs["last_v_r"], arrs["last_v_c"], arrs["last_dones"], *params)
    for W in (0, 1):
        adv_r, adv_c, ret_r, ret_c = _run(arrs, params, W)
        for got, key in ((adv_r, "reward_advantages"), (adv_c, "cost_advantages"), (ret_r, "reward_returns"), (ret_c, "cost_returns")):
            if W == 1 or N >= 65536:
                assert np.array_equal(got, o[key]), (key, W)
            else:
                assert np.allclose(got, o[key], rtol=2e-7, atol=1e-7), (key, W)
                assert (got != o[key]).mean() < 1e-5


def test_gae_linearity_full_size():
    """Size-independent property at BASELINE scale (N = 65536, T = 2048 would be 4.8 GB of inputs; use 16384 x 2048):
    with dones = 0, GAE is linear in (rewards, values): gae(x + y) == gae(x) + gae(y) to float32 rounding."""
    T, N = 2048, 16384
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    from icrl_amd import _lib
    L = _lib.lib()

    def run(r, v):
        z = torch.zeros(T, N, device=dev)
        outs = [torch.empty(T, N, device=dev) for _ in range(4)]
        ld = torch.zeros(N, dtype=torch.uint8, device=dev)
        _lib.check(L.icrl_gae_dual(*(_lib.ptr(x) for x in (r, r, v, v, z, v[-1].contiguous(), v[-1].contiguous(), ld, *outs)),
                                   T, N, 0.99, 0.95, 0.99, 0.95, _lib.current_stream()), "gae")
        return outs[0], outs[1]
    r1, v1 = torch.randn(T, N, device=dev, generator=g), torch.randn(T, N, device=dev, generator=g)
    r2, v2 = torch.randn(T, N, device=dev, generator=g), torch.randn(T, N, device=dev, generator=g)
    a1, c1 = run(r1, v1)
    a2, _ = run(r2, v2)
    a12, _ = run(r1 + r2, v1 + v2)
    assert torch.equal(a1, c1)                         # reward and cost chains fed the same data agree bit-for-bit
    assert torch.allclose(a12, a1 + a2, rtol=1e-4, atol=1e-4)


def _run_ws(arrs, params, shape, ws):
    from icrl_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    f = lambda k: torch.as_tensor(np.ascontiguousarray(arrs[k], dtype=np.float32), device=dev)
    r, c, vr, vc, d = f("rewards"), f("costs"), f("reward_values"), f("cost_values"), f("dones")
    lvr, lvc = f("last_v_r"), f("last_v_c")
    ld = torch.as_tensor(np.asarray(arrs["last_dones"]).astype(np.uint8), device=dev)
    T, N = r.shape
    outs = [torch.full((T, N), float("nan"), device=dev) for _ in range(4)]
    _lib.check(L.icrl_gae_dual_ws(*(_lib.ptr(x) for x in (r, c, vr, vc, d, lvr, lvc, ld, *outs)), T, N, *[float(p) for p in params],
                                  shape, _lib.ptr(ws), ws.numel() * 8, _lib.current_stream()), "icrl_gae_dual_ws")
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs]


def _ws(T=2048, N=1024):
    """a zeroed workspace of the size the library asks for at this shape (never below ICRL_GAE_WS_BYTES)."""
    from icrl_amd import _lib
    from icrl_amd.structs import GAE_WS_BYTES
    return torch.zeros(max(int(_lib.lib().icrl_gae_dual_ws_bytes(T, N)), GAE_WS_BYTES) // 8, dtype=torch.int64, device="cuda:0")


@pytest.mark.parametrize("case", ["t8n3", "t2000n1", "t256n16", "t1n4", "t64n5"])
@pytest.mark.parametrize("C", [2, 3, 16])
def test_gae_split_over_workgroups_golden(golden, case, C):
    """two-level scan (time axis split over C workgroups x 8 waves): the reference's golden vectors, incl. chunks that are empty
    (T = 1, T = 8 with 16 x 8 waves) and ragged; one workspace reused by every launch."""
    g = golden("g1_gae")
    arrs = {k.split("/")[1]: g[k] for k in g.files if k.startswith(case + "/")}
    ws = _ws()
    for _ in range(2):
        outs = _run_ws(arrs, arrs["params"], 200 + C, ws)
        for got, key in zip(outs, ("reward_advantages", "cost_advantages", "reward_returns", "cost_returns")):
            assert np.allclose(got, arrs[key], rtol=2e-7, atol=1e-7), key


@pytest.mark.parametrize("T,N", [(2048, 64), (2048, 256), (512, 256), (256, 512), (2048, 4096), (1000, 130), (100, 64)])
def test_gae_split_vs_oracle_random(T, N):
    """the shapes the loop launches (64 envs x 2048, the shard shapes of configs[2..4]) through the default heuristic with a
    workspace: <= 1 ulp of float32 from the sequential scan, and almost everywhere equal."""
    rng = np.random.RandomState(T + N)
    arrs = dict(rewards=rng.randn(T, N), costs=rng.rand(T, N), reward_values=rng.randn(T, N), cost_values=rng.randn(T, N),
                dones=(rng.rand(T, N) < 0.002), last_v_r=rng.randn(N), last_v_c=rng.randn(N), last_dones=rng.rand(N) < 0.2)
    arrs = {k: (v.astype(np.float32) if v.dtype != bool else v) for k, v in arrs.items()}
    params = (0.99, 0.95, 0.99, 0.9)
    o = o_gae.dual_gae(arrs["rewards"], arrs["costs"], arrs["reward_values"], arrs["cost_values"],
                       arrs["dones"].astype(np.float32), arrs["last_v_r"], arrs["last_v_c"], arrs["last_dones"], *params)
    ws = _ws(T, N)
    for rep in range(3):
        outs = _run_ws(arrs, params, 0, ws)
        for got, key in zip(outs, ("reward_advantages", "cost_advantages", "reward_returns", "cost_returns")):
            assert np.allclose(got, o[key], rtol=2e-7, atol=1e-7), (key, rep)
            assert (got != o[key]).mean() < 1e-5


def _random_arrs(T, N, seed=None):
    rng = np.random.RandomState(T + N if seed is None else seed)
    arrs = dict(rewards=rng.randn(T, N), costs=rng.rand(T, N), reward_values=rng.randn(T, N), cost_values=rng.randn(T, N),
                dones=(rng.rand(T, N) < 0.002), last_v_r=rng.randn(N), last_v_c=rng.randn(N), last_dones=rng.rand(N) < 0.2)
    return {k: (v.astype(np.float32) if v.dtype != bool else v) for k, v in arrs.items()}


@pytest.mark.parametrize("case", ["t8n3", "t2000n1", "t256n16", "t1n4", "t64n5"])
def test_gae_register_resident_split_golden(golden, case):
    """the register-resident split scan (round 6; shape code 500) on the reference's golden vectors: T = 1 (head row only), T = 8 (one wave),
    T = 2000 (16 chunks, the last ragged), one workspace reused by consecutive launches."""
    g = golden("g1_gae")
    arrs = {k.split("/")[1]: g[k] for k in g.files if k.startswith(case + "/")}
    T, N = arrs["rewards"].shape
    ws = _ws(T, N)
    for _ in range(2):
        outs = _run_ws(arrs, arrs["params"], 500, ws)
        for got, key in zip(outs, ("reward_advantages", "cost_advantages", "reward_returns", "cost_returns")):
            assert np.allclose(got, arrs[key], rtol=2e-7, atol=1e-7), key
    assert int(ws.view(torch.int32)[-1].item()) == 0


@pytest.mark.parametrize("T,N", [(2048, 8192), (1024, 32768), (1000, 16390), (129, 9000), (256, 65472)])
def test_gae_mid_range_vs_oracle(T, N):
    """8 192 .. 65 472 envs (between the cache-resident and the streaming regime) through the DEFAULT heuristic with the
    workspace the library asks for: 2 048 .. 16 368 workgroups of the register-resident split scan, non-temporal accesses; ragged tiles and
    ragged chunks; <= 1 ulp of float32 from the sequential scan and almost everywhere equal (ref: buffers.py:493-552)."""
    arrs = _random_arrs(T, N)
    params = (0.99, 0.95, 0.99, 0.9)
    o = o_gae.dual_gae(arrs["rewards"], arrs["costs"], arrs["reward_values"], arrs["cost_values"],
                       arrs["dones"].astype(np.float32), arrs["last_v_r"], arrs["last_v_c"], arrs["last_dones"], *params)
    ws = _ws(T, N)
    for rep in range(2):
        outs = _run_ws(arrs, params, 0, ws)
        for got, key in zip(outs, ("reward_advantages", "cost_advantages", "reward_returns", "cost_returns")):
            assert np.allclose(got, o[key], rtol=2e-7, atol=1e-7), (key, rep)
            assert (got != o[key]).mean() < 1e-5
    assert int(ws.view(torch.int32)[-1].item()) == 0
    # and it IS the split scan that ran: the same launch with the sequential shape forced is bit-exact, this one is not required to be
    seq = _run_ws(arrs, params, 1, ws)
    assert np.array_equal(seq[0], o["reward_advantages"])


@pytest.mark.parametrize("shape", [501, 304])
def test_gae_split_wait_is_bounded_and_reported(shape):
    """every wait of the split scans ends.  Shape codes 501 (register-resident) / 300 + C (two-pass) withhold the map of the
    workgroup holding the LATEST rows and shorten the poll limit: the launch must end, the workspace's status word must be 1, and
    RolloutBufferWithCost.check_gae_status must raise and clear it; the next healthy launch on the same workspace is correct again."""
    T, N = 1024, 192
    arrs = _random_arrs(T, N, 7)
    params = (0.99, 0.95, 0.99, 0.95)
    ws = _ws(T, N)
    _run_ws(arrs, params, shape, ws)                      # returns (synchronised) instead of hanging
    status = ws.view(torch.int32)[-1:]
    assert int(status.item()) == 1
    from icrl_amd.buffers import RolloutBufferWithCost
    rb = RolloutBufferWithCost.__new__(RolloutBufferWithCost)
    rb.gae_status = status
    with pytest.raises(RuntimeError, match="affine map"):
        rb.check_gae_status()
    assert int(status.item()) == 0
    o = o_gae.dual_gae(arrs["rewards"], arrs["costs"], arrs["reward_values"], arrs["cost_values"],
                       arrs["dones"].astype(np.float32), arrs["last_v_r"], arrs["last_v_c"], arrs["last_dones"], *params)
    outs = _run_ws(arrs, params, 500 if shape == 501 else 204, ws)
    assert np.allclose(outs[0], o["reward_advantages"], rtol=2e-7, atol=1e-7) and int(status.item()) == 0
