"""Several independent ICRL runs (seeds) sharing ONE GPU.

One run occupies 3 CUs during its PPO-Lagrangian update (the three persistent workgroups of icrl_ppo_lag_train) and one CU
per environment during a rollout — 1-25 % of an MI355X.  north_star: "independent seeds ... fan out"; on one GPU they fan out
over HIP streams: run s has its own stream, its own host thread (the C-ABI calls and the blocking statistics read-backs
release the GIL) and its own random streams (icrl_amd/streams.py), so every run computes exactly what it computes alone
(tests/test_seed_batch_gpu.py: bit-identical parameters) while the GPU executes the runs' kernels side by side.

The persistent kernels assume all their workgroups are co-resident (they wait for each other inside the launch); two of them
fit side by side only while the CUs last.  CuBudget is the host-side admission control: a launch takes one token per
workgroup (each of these workgroups fills a CU's LDS) and gives them back when it has completed.
"""
import threading
import time
import types

import torch

from . import _lib, logger
from .streams import PrivateStreams


class CuBudget:
    def __init__(self, total):
        self.total, self.free, self.cv = int(total), int(total), threading.Condition()

    def acquire(self, n):
        n = min(int(n), self.total)
        with self.cv:
            while self.free < n:
                self.cv.wait()
            self.free -= n
        return n

    def release(self, n):
        with self.cv:
            self.free += n
            self.cv.notify_all()


class budgeted:
    """`with budgeted(n_workgroups):` around a persistent launch; a no-op when no budget is installed (single run)."""

    def __init__(self, n):
        self.n, self.b = n, _lib.CU_BUDGET

    def __enter__(self):
        if self.b is not None:
            self.n = self.b.acquire(self.n)

    def __exit__(self, *exc):
        if self.b is not None:
            torch.cuda.current_stream().synchronize()      # the launch has left the CUs
            self.b.release(self.n)


def setup_runs(configs, on_setup=None):
    """icrl.setup() for every config, one after the other (the constructors seed the process-wide generators); each run gets
    its own PrivateStreams unless the config already carries a `streams` object."""
    from . import icrl as I
    states = []
    for cfg in configs:
        if getattr(cfg, "streams", None) is None:
            cfg.streams = PrivateStreams(cfg.seed, discrete=cfg.train_env_id in ("LGW-v0", "CLGW-v0"))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st = I.setup(cfg)
            if on_setup is not None:
                on_setup(st)
        st["stream"] = s
        states.append(st)
    torch.cuda.synchronize()
    return states


def run_iterations(states, first_iteration, n_iters):
    """outer iterations [first_iteration, first_iteration + n_iters) of every run, concurrently: one host thread and one HIP
    stream per run.  Returns (metrics per run, wall seconds).  ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4)
    hardware queues: export GPU_MAX_HW_QUEUES >= len(states) before the process first touches the GPU, or at most 4 runs
    overlap (measured: 2.2 M aggregate env-steps/s at 4 queues, 5.7 M at 32, 32 runs of BASELINE configs[1])."""
    from . import icrl as I
    n_cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    dev = torch.cuda.current_device()
    _lib.CU_BUDGET = CuBudget(n_cus) if len(states) > 1 else None
    out, errors = [[] for _ in states], []

    def worker(i):
        try:
            torch.cuda.set_device(dev)
            logger.configure()               # the scalar log is per thread
            with torch.cuda.stream(states[i]["stream"]):
                for it in range(first_iteration, first_iteration + n_iters):
                    out[i].append(I.outer_iteration(states[i], it))
                states[i]["stream"].synchronize()
        except BaseException as e:           # noqa: BLE001 - reported to the caller below
            errors.append((i, e))

    torch.cuda.synchronize()
    t0 = time.time()
    try:
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(states))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
    finally:
        _lib.CU_BUDGET = None
    dt = time.time() - t0
    if errors:
        raise RuntimeError(f"run {errors[0][0]} failed: {errors[0][1]!r}") from errors[0][1]
    return out, dt


def run_seed_batch(configs, n_iters, on_setup=None):
    """configs: one icrl config (types.SimpleNamespace, see icrl.build_parser) per run.  Runs `n_iters` outer iterations of
    every run concurrently.  Returns (states, metrics per run, wall seconds of the iteration phase)."""
    states = setup_runs(configs, on_setup)
    out, dt = run_iterations(states, 0, n_iters)
    return states, out, dt
