import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icrl_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); T = 2048
for N in tuple(int(x) for x in os.environ.get('NS', '65536,131072').split(',')):
    ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.001).float()]
    lv = [torch.randn(N, device=dev) for _ in range(2)]; ld = torch.zeros(N, dtype=torch.uint8, device=dev)
    outs = [torch.empty(T, N, device=dev) for _ in range(4)]
    args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
    st = _lib.current_stream()
    res = {}
    for rnd in range(3):           # interleaved rounds in one process
        for W in (106, 111, 112):
            L.icrl_gae_dual_ex(*args, T, N, 0.99, 0.95, 0.99, 0.95, W, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): L.icrl_gae_dual_ex(*args, T, N, 0.99, 0.95, 0.99, 0.95, W, st)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(W, []).append(T * N * 36 / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e9)
    print(N, {W: f"{min(v):.0f}-{max(v):.0f}" for W, v in res.items()})
    del ins, outs
# the four-column shapes are bit-exact replicas of the sequential scan
T, N = 300, 4096
ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.01).float()]
lv = [torch.randn(N, device=dev) for _ in range(2)]; ld = (torch.rand(N, device=dev) < 0.3).to(torch.uint8)
ref = [torch.empty(T, N, device=dev) for _ in range(4)]
L.icrl_gae_dual_ex(*[_lib.ptr(x) for x in (*ins, *lv, ld, *ref)], T, N, 0.99, 0.95, 0.99, 0.9, 1, st)
for W in (107, 108, 109, 110, 111, 112):
    outs = [torch.full((T, N), float("nan"), device=dev) for _ in range(4)]
    L.icrl_gae_dual_ex(*[_lib.ptr(x) for x in (*ins, *lv, ld, *outs)], T, N, 0.99, 0.95, 0.99, 0.9, W, st)
    torch.cuda.synchronize()
    print(W, "bit-exact" if all(torch.equal(a, b) for a, b in zip(outs, ref)) else "DIFFERS")
