"""one-size GAE launch loop for PMC collection (FETCH_SIZE / WRITE_SIZE per dispatch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icrl_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
T, N = 2048, int(os.environ.get("GAE_N", "131072"))
ins = [torch.randn(T, N, device=dev) for _ in range(4)] + [(torch.rand(T, N, device=dev) < 0.001).float()]
lv = [torch.randn(N, device=dev) for _ in range(2)]; ld = torch.zeros(N, dtype=torch.uint8, device=dev)
outs = [torch.empty(T, N, device=dev) for _ in range(4)]
args = [_lib.ptr(x) for x in (*ins, *lv, ld, *outs)]
for _ in range(4):
    L.icrl_gae_dual(*args, T, N, 0.99, 0.95, 0.99, 0.95, _lib.current_stream())
torch.cuda.synchronize()
print("algorithmic bytes per launch", T * N * 36)
