"""Config-size GAE launches (the shapes the loop itself runs): one workgroup per column tile vs the time axis split over
workgroups (icrl_gae_dual_ws).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icrl_amd import _lib
from icrl_amd.structs import GAE_WS_BYTES
L = _lib.lib()
dev = torch.device("cuda:0")
ws = torch.zeros(GAE_WS_BYTES // 8 + 1, dtype=torch.int64, device=dev)
for T, N in ((2048, 64), (1024, 256), (512, 256), (256, 512), (2048, 4096)):
    ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.002).float()]
    lv = [torch.randn(N, device=dev), torch.randn(N, device=dev), torch.zeros(N, dtype=torch.uint8, device=dev)]
    outs = [torch.empty(T, N, device=dev) for _ in range(4)]
    for shape, use_ws in ((16, False), (0, False), (0, True), (204, True), (208, True), (216, True)):
        args = [_lib.ptr(x) for x in (*ins, *lv, *outs)] + [T, N, 0.99, 0.95, 0.99, 0.95, shape, _lib.ptr(ws) if use_ws else None, ws.numel() * 8 if use_ws else 0, _lib.current_stream()]
        err = L.icrl_gae_dual_ws(*args)
        if err:
            L.icrl_clear_error(); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.icrl_gae_dual_ws(*args)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"T={T} N={N} shape={shape:3d} ws={int(use_ws)}: {us:7.1f} us/launch = {36 * T * N / us / 1e3:8.1f} GB/s")
