"""Per-run random streams (host side).

By default a run draws from the process-wide generators exactly like the reference (torch for action noise, numpy for
minibatch permutations, common/utils.py:23-39).  When several independent runs share one process (icrl_amd/seed_batch.py)
each needs generators of its own, otherwise the runs' draws interleave and no run is reproducible: PrivateStreams answers the
`streams` protocol of PPOLagrangian / icrl.outer_iteration (rollout_noise, permutation, consumed, sample_noise, eval_noise —
the protocol the parity tests use for teacher forcing, oracle/streams.py) from a private device generator seeded with the run's seed.

Minibatch permutations are drawn ON THE DEVICE here (torch.randperm with the private generator) instead of with
np.random.permutation: 20 permutations of 131 072 entries per outer iteration cost ~20 ms of host time each way, and with dozens
of runs in one process the host threads serialise on them (measured: 32 runs spent 1.3 of 1.46 s per iteration there).  Any
uniformly random permutation is a valid minibatch order (ref: buffers.py:596); parity tests force the order anyway.
"""
import numpy as np
import torch


class PrivateStreams:
    def __init__(self, seed, device="cuda", discrete=False):
        self.device, self.discrete = torch.device(device), discrete
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        self.perm_gen = torch.Generator(device=self.device)      # its own generator: the number of executed epochs (early stop)
        self.perm_gen.manual_seed(int(seed) + 0x5eed)            # must not shift the action-noise stream

    def _draw(self, *shape):
        if self.discrete:
            return torch.rand(shape[:-1], device=self.device, generator=self.gen)
        return torch.randn(shape, device=self.device, generator=self.gen)

    def rollout_noise(self, T, N, A):
        return self._draw(T, N, A)

    def permutation(self, epoch, n):
        return torch.randperm(n, device=self.device, generator=self.perm_gen)

    def permutations(self, n_epochs, n):
        """all epochs' permutations of one train() call in ONE batched draw: [n_epochs, n] = the sort order of n_epochs x n float64
        uniforms (no ties in practice: 2^53 values).  torch.randperm per epoch is a sort per call — with dozens of runs in a batch
        the draws of one update phase were 20+ ms of device time between the rollout and the update launch."""
        pre = getattr(self, "_prefetched", None)
        if pre is not None:
            self._prefetched = None
            if pre[0] == (n_epochs, n):             # drawn ahead on a side stream (prefetch_permutations): the same draw, earlier
                torch.cuda.current_stream().wait_event(pre[2])
                pre[1].record_stream(torch.cuda.current_stream())
                return pre[1]
            raise RuntimeError("PrivateStreams: permutations were prefetched for another shape")      # (would desynchronise the stream)
        keys = torch.rand(n_epochs, n, dtype=torch.float64, device=self.device, generator=self.perm_gen)
        return torch.argsort(keys, dim=1).to(torch.int32)

    def prefetch_permutations(self, n_epochs, n, stream):
        """draw the NEXT permutations() call's result now, on `stream` — e.g. while an update occupies 3 CUs and the host waits for
        it.  The draw consumes the permutation generator exactly as the later call would have (nothing else may draw from it in
        between: the caller's business), so the values are the same; only when the sorts run changes."""
        if getattr(self, "_prefetched", None) is not None:
            return
        with torch.cuda.stream(stream):
            keys = torch.rand(n_epochs, n, dtype=torch.float64, device=self.device, generator=self.perm_gen)
            perms = torch.argsort(keys, dim=1).to(torch.int32)
            ev = torch.cuda.Event()
            ev.record(stream)
        self._prefetched = ((n_epochs, n), perms, ev)

    def consumed(self, executed_epochs):
        pass

    def cn_permutations(self, iterations, size):
        """the np.random.permutation draws of ConstraintNet.get() in minibatch mode (`--cn_batch_size`, constraint_net.py:300-316):
        [iterations, size] from this run's own generator (the process-wide numpy generator would interleave the runs' draws)."""
        assert getattr(self, "_prefetched", None) is None, "update permutations were drawn ahead across a constraint-net draw"
        return torch.stack([torch.randperm(size, device=self.device, generator=self.perm_gen) for _ in range(max(int(iterations), 1))])

    def sample_noise(self, rows, A):
        return self._draw(rows, A)

    def eval_noise(self, rows, A):
        return self._draw(rows, A)
