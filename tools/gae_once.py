"""one-size GAE launch loop for PMC collection (FETCH_SIZE / WRITE_SIZE per dispatch); GAE_N = envs (default 131 072: the streaming shape;
32 768: the register-resident split scan), through icrl_gae_dual_ws with the workspace the library asks for, as RolloutBufferWithCost calls it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icrl_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
T, N = 2048, int(os.environ.get("GAE_N", "131072"))
ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.001).float()]
lv = [torch.randn(N, device=dev) for _ in range(2)]; ld = torch.zeros(N, dtype=torch.uint8, device=dev)
outs = [torch.empty(T, N, device=dev) for _ in range(4)]
ws = torch.zeros(int(L.icrl_gae_dual_ws_bytes(T, N)) // 8, dtype=torch.int64, device=dev)
args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
for _ in range(4):
    L.icrl_gae_dual_ws(*args, T, N, 0.99, 0.95, 0.99, 0.95, 0, _lib.ptr(ws), ws.numel() * 8, _lib.current_stream())
torch.cuda.synchronize()
assert int(ws.view(torch.int32)[-1].item()) == 0
print("algorithmic bytes per launch", T * N * 36)
