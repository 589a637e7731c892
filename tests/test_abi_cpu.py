"""CPU: the C-ABI library loads and exports every symbol include/icrl_hip.h declares (no compute without a GPU);
host-side logic that needs no kernel."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "icrl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icrl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from icrl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/icrl_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert L.icrl_abi_version() == 106
    assert L.icrl_cn_train_work_floats(521, 10000, 5000, 10) > 521 * 235


def test_refused_arguments_carry_a_reason():
    """hard limits of the kernels are reported with text, not a bare hipErrorInvalidValue (argument checks run on the host,
    before any launch: no GPU needed)."""
    import pytest
    from icrl_amd import _lib, structs as S
    L = _lib.lib()
    pol = S.PolicyT(18, 6, 32, 64, 0, 1, None, None)                    # hidden (32, 64): the update kernels are 64 x 64
    buf = S.BufferT(8, 4, 18, 0)
    hp = S.PpoHyperT(64, 2, 0, 0)
    err = L.icrl_ppo_lag_train(ctypes.byref(pol), None, None, None, ctypes.byref(buf), None, None, ctypes.byref(hp), None, None, None)
    assert err == 1
    with pytest.raises(ValueError, match=r"hidden widths \(32, 64\).*64 x 64"):
        _lib.check(err, "icrl_ppo_lag_train")
    assert L.icrl_last_error() == b""                                  # consumed by check()
    err = L.icrl_gae_dual(*([None] * 12), 0, 5, 0.99, 0.95, 0.99, 0.95, None)
    with pytest.raises(ValueError, match="T = 0, N = 5"):
        _lib.check(err, "icrl_gae_dual")
    # batch sizes above 256 and layers above 64 are served (generic-shape path); what it refuses says so
    hp = S.PpoHyperT(512, 2, 0, 0)
    pol = S.PolicyT(18, 6, 320, 320, 0, 1, None, None)
    err = L.icrl_ppo_lag_train(ctypes.byref(pol), None, None, None, ctypes.byref(buf), None, None, ctypes.byref(hp), None, None, None)
    with pytest.raises(ValueError, match=r"hidden widths \(320, 320\).*multiple of 64 up to 256"):
        _lib.check(err, "icrl_ppo_lag_train")
    hp = S.PpoHyperT(1, 2, 0, 0)
    pol = S.PolicyT(18, 6, 128, 128, 0, 1, None, None)
    err = L.icrl_ppo_lag_train(ctypes.byref(pol), None, None, None, ctypes.byref(buf), None, None, ctypes.byref(hp), None, None, None)
    with pytest.raises(ValueError, match="batch_size 1"):
        _lib.check(err, "icrl_ppo_lag_train")


def test_generic_shape_limits_are_reported():
    """what the generic-shape path refuses (host arithmetic, before any launch): more than 4 layers per group / 256 units per layer in an
    `arch` descriptor, constraint nets outside 0..4 hidden layers or beyond the LDS; icrl_ppo_generic_row_floats sizes the scratch."""
    from icrl_amd import _lib, structs as S
    L = _lib.lib()
    def pol(desc, n_params=1):
        a = np.ascontiguousarray(desc, dtype=np.int32)
        return S.PolicyT(18, 6, 0, 0, 0, n_params, None, None, a.ctypes.data), a
    p, keep = pol([1, 48, 3, 64, 32, 32, 1, 40, 0])       # tests/helpers/arches.py "trunk": -sl 48 -pl 64 32 32 -rvl 40 -cvl
    n = 6 + (18 * 48 + 48) + (48 * 64 + 64) + (64 * 32 + 32) + (32 * 32 + 32) + (48 * 40 + 40) + (32 * 6 + 6) + (40 + 1) + (48 + 1)
    p.n_params = n
    assert L.icrl_ppo_generic_row_floats(ctypes.byref(p)) == 48 + 64 + 32 + 32 + 40 + 6 + 2
    p.n_params = n + 1
    assert L.icrl_ppo_generic_row_floats(ctypes.byref(p)) == -1
    assert f"the architecture needs {n}".encode() in L.icrl_last_error()
    p, keep = pol([5, 8, 8, 8, 8, 8, 0, 0, 0])
    assert L.icrl_ppo_generic_row_floats(ctypes.byref(p)) == -1 and b"5 shared layers (0..4)" in L.icrl_last_error()
    p, keep = pol([0, 1, 300, 0, 0])
    assert L.icrl_ppo_generic_row_floats(ctypes.byref(p)) == -1 and b"300 units (1..256)" in L.icrl_last_error()
    cn = S.CostNetT(18, 6, 24, 5, 8, 8, 8, 8, 0, 1)
    err = L.icrl_cost_mlp_forward(ctypes.byref(cn), None, None, 4, None, None)
    with pytest.raises(ValueError, match=r"5 hidden layers \(0\.\.4\)"):
        _lib.check(err, "icrl_cost_mlp_forward")
    n_cn = (24 * 128 + 128) + 3 * (128 * 128 + 128) + 129
    cn = S.CostNetT(18, 6, 24, 4, 128, 128, 128, 128, 0, n_cn)
    err = L.icrl_cost_mlp_forward(ctypes.byref(cn), None, None, 4, None, None)
    with pytest.raises(ValueError, match="B of LDS, 160 KB available"):
        _lib.check(err, "icrl_cost_mlp_forward")


def test_struct_sizes_match_header_layout():
    """natural alignment, no packing: sizes computed by hand from include/icrl_hip.h."""
    from icrl_amd import structs as S
    assert ctypes.sizeof(S.EnvT) == 8 * 4 + 5 * 8
    assert ctypes.sizeof(S.NormT) == 4 * 4 + 6 * 8 + 7 * 8
    assert ctypes.sizeof(S.PolicyT) == 6 * 4 + 3 * 8
    assert ctypes.sizeof(S.CostNetT) == 10 * 4 + 8 + 5 * 8 + 8 + 2 * 8
    assert ctypes.sizeof(S.BufferT) == 4 * 4 + 18 * 8
    assert ctypes.sizeof(S.AgentT) == 11 * 8
    assert ctypes.sizeof(S.PpoHyperT) == 4 * 4 + 12 * 4
    assert ctypes.sizeof(S.CnHyperT) == 4 * 4 + 8 * 4


def test_missing_library_fails_loudly(monkeypatch):
    from icrl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libicrl_hip.so")
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "icrl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_dual_variable_matches_reference_trajectories(golden):
    """host-side Lagrange multiplier (numpy float32) vs the reference's DualVariable (tests/golden/g5)."""
    from icrl_amd.dual_variable import DualVariable
    g = golden("g5_dual")
    for case in "abcd":
        nu0, lr, budget = (float(x) for x in g[case + "/params"])
        d = DualVariable(budget, lr, nu0, None)
        for c, (nu, loss, log_nu) in zip(g[case + "/costs"], g[case + "/traj"]):
            d.update_parameter(c)
            assert abs(d.nu().item() - nu) <= 2e-6 * max(1.0, abs(nu))


def test_flag_set_matches_reference_readme_commands():
    from icrl_amd.icrl import build_parser
    p = build_parser()
    a = p.parse_args("icrl -p ICRL-FE2 --group HC-ICRL -er 10 -ep x -tk 0.01 -cl 20 -bi 10 -ft 2e5 -ni 30 -tei HCWithPos-v0 "
                     "-eei HCWithPosTest-v0 -clr 0.05 -aclr 0.9 -crc 0.5 -psis -ctkno 2.5".split())
    assert a.forward_timesteps == 200000 and a.cn_layers == [20] and a.per_step_importance_sampling and a.cn_target_kl_new_old == 2.5
    a = p.parse_args("icrl -ep x -er 45 -cl 40 40 -clr 0.005 -aclr 0.9 -crc 0.6 -bi 5 -ft 2e5 -ni 20 -tei AntWall-v0 -eei AntWallTest-v0 "
                     "--batch_size 128 --reward_gae_lambda 0.9 --cost_gae_lambda 0.9 --n_epochs 20 --learning_rate 3e-5 "
                     "--clip_range 0.4 -piv 0.1 -plr 0.05 -psis -tk 0.02 -ctkno 2.5".split())
    assert a.cn_layers == [40, 40] and a.batch_size == 128 and a.penalty_initial_value == 0.1 and a.n_epochs == 20


def test_gpu_suite_fits_its_time_budget():
    """tests/gpu_suite_duration.json = the record the last FULL `pytest tests -m gpu` run on an MI355X box left (tests/conftest.py writes
    it to gpurun_out/, it is committed from there): the driver's step limit for that command is 1 200 s, the budget 900 s."""
    import json
    from conftest import GPU_SUITE_BUDGET_S
    rec = json.load(open(os.path.join(ROOT, "tests", "gpu_suite_duration.json")))
    assert rec["tests"] >= 266, "the record must come from a run of the whole GPU suite"
    assert rec["duration_s"] <= GPU_SUITE_BUDGET_S, f"GPU suite {rec['duration_s']} s > {GPU_SUITE_BUDGET_S} s: slowest {rec['slowest']}"
