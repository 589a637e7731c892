import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from icrl_amd import icrl as I, _lib
cfg = bench.config2(4, 0, 0, 1)
st = I.setup(cfg)
I.outer_iteration(st, 0)
ag = st["agent"]
L = _lib.lib()
orig = L.icrl_ppo_lag_train
def timed(*a):
    t0 = time.perf_counter(); r = orig(*a); print("icrl_ppo_lag_train host ms", 1e3 * (time.perf_counter() - t0)); return r
L.icrl_ppo_lag_train = timed
for name in ("prepare",):
    f = getattr(ag.policy, name)
    def g(*a, _f=f, _n=name, **k):
        t0 = time.perf_counter(); r = _f(*a, **k); print(_n, "host ms", 1e3 * (time.perf_counter() - t0)); return r
    setattr(ag.policy, name, g)
f2 = ag._draw_permutations
def g2(n):
    t0 = time.perf_counter(); r = f2(n); print("perms host ms", 1e3 * (time.perf_counter() - t0)); return r
ag._draw_permutations = g2
for i in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ag.train()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("train() ms", 1e3 * (t1 - t0))
