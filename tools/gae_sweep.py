"""GAE kernel sweep on the GPU: achieved algorithmic GB/s (36 B/transition) vs N and launch shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icrl_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
T = 2048
print("| envs N (T = 2048) | algorithmic MB | us / launch | GB/s | fraction of 8 TB/s |\n|---|---|---|---|---|")
for N in [64, 256, 512, 2048, 8192, 32768, 65536, 131072]:
    ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.001).float()]
    lv = [torch.randn(N, device=dev) for _ in range(2)]
    ld = torch.zeros(N, dtype=torch.uint8, device=dev)
    outs = [torch.empty(T, N, device=dev) for _ in range(4)]
    args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
    for W in (0,):       # 0 = the library's own choice of launch shape (what the loop uses)
        st = _lib.current_stream()
        for _ in range(3):
            L.icrl_gae_dual_ex(*args, T, N, 0.99, 0.95, 0.99, 0.95, W, st)
        torch.cuda.synchronize()
        reps = 20 if N <= 4096 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.icrl_gae_dual_ex(*args, T, N, 0.99, 0.95, 0.99, 0.95, W, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gb = T * N * 36 / 1e9
        print(f"| {N} | {gb*1e3:.1f} | {ms*1e3:.1f} | {gb/ (ms/1e3):.0f} | {gb/ (ms/1e3)/8000:.3f} |")
    del ins, outs
