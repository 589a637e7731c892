"""GAE launch shapes between the cache-resident and the streaming regime: GB/s (36 B per transition) of icrl_gae_dual_ws at
NS envs x 2048 rows for the shape codes in CODES — 0: the library's choice (the register-resident split scan below 65 536 envs), 500: that scan
forced, 4 / 101 / 106: the shapes the heuristic used until round 5 (4 waves per tile through LDS, one wave per tile).  Interleaved rounds, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icrl_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); T = int(os.environ.get("T", "2048"))
codes = tuple(int(x) for x in os.environ.get("CODES", "0,500,4,101,106").split(","))
print("| envs | " + " | ".join(f"code {c}" for c in codes) + " |\n|---|" + "---|" * len(codes))
for N in tuple(int(x) for x in os.environ.get("NS", "4096,8192,16384,32768,65536").split(",")):
    ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.001).float()]
    lv = [torch.randn(N, device=dev) for _ in range(2)]; ld = torch.zeros(N, dtype=torch.uint8, device=dev)
    outs = [torch.empty(T, N, device=dev) for _ in range(4)]
    tiles = (N + 63) // 64
    ws = torch.zeros(max(int(L.icrl_gae_dual_ws_bytes(T, N)), tiles * 32 * 2052 + 64) // 8 + 1, dtype=torch.int64, device=dev)
    args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
    st = _lib.current_stream()
    res = {}
    for rnd in range(3):
        for W in codes:
            call = lambda: L.icrl_gae_dual_ws(*args, T, N, 0.99, 0.95, 0.99, 0.95, W, _lib.ptr(ws), ws.numel() * 8, st)
            if call() != 0:
                res[W] = None; L.icrl_clear_error(); continue
            torch.cuda.synchronize()
            reps = 5 if N >= 32768 else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): call()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(W, []).append(T * N * 36 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9)
    print(f"| {N} | " + " | ".join("-" if res[W] is None else f"{min(res[W]):.0f}-{max(res[W]):.0f}" for W in codes) + " |", flush=True)
    del ins, outs
