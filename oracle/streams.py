"""Oracle: the random streams of one ICRL run made explicit, so that the CPU port, the HIP path and the reference can be
driven by the SAME draws (teacher forcing).  Test infrastructure only.

The reference consumes, in this order per outer iteration (icrl/icrl.py:199-252):
  per rollout   n_steps action draws [N, act] (Normal.rsample / Categorical.sample, on_policy_algorithm.py:367-372)
  per train()   one np.random.permutation(T*N) per EXECUTED epoch (buffers.py:596; none after a target-KL stop)
  sampling      one action draw per step of `expert_rollouts` episodes (icrl/utils.py:323-357)
  evaluation    one action draw per step of 10 episodes (common/evaluation.py:10-67)

A stream object answers
  rollout_noise(T, N, A) -> [T, N, A] standard normals (continuous) or [T, N] numbers u in [0, 1] (discrete: action =
                            #{k : u >= cdf_k}; recorded actions a are replayed as u = float(a) for two classes)
  permutation(epoch, n)  -> permutation `epoch` of the current train() call;  consumed(k) closes the call after k epochs
  sample_noise(rows, A), eval_noise(rows, A) -> [rows, A] (or [rows]) draws for the 1-env episode loops
"""
import numpy as np


class SeededStreams:
    """draws keyed by (kind, call index): independent of how many values an implementation happens to request."""

    def __init__(self, seed, discrete=False, uniform=False):
        # uniform: draws in [0, 1) with the continuous shapes ([T, N, 1] for a Discrete action space): the inverse-CDF uniforms of a Categorical policy
        # (standard normals used as "uniforms" force the action whenever they fall outside [0, 1], i.e. in 58 % of the steps)
        self.seed, self.discrete, self.uniform = int(seed), discrete, uniform
        self.calls = dict(rollout=0, train=0, sample=0, eval=0)

    def _rng(self, kind, *idx):
        return np.random.RandomState([self.seed, dict(rollout=1, train=2, sample=3, eval=4)[kind], *idx])

    def _draw(self, rng, shape):
        if self.uniform:
            return rng.rand(*shape).astype(np.float32)
        return rng.rand(*shape[:-1]).astype(np.float32) if self.discrete else rng.randn(*shape).astype(np.float32)

    def rollout_noise(self, T, N, A):
        k = self.calls["rollout"]; self.calls["rollout"] += 1
        return self._draw(self._rng("rollout", k), (T, N, A))

    def permutation(self, epoch, n):
        return self._rng("train", self.calls["train"], epoch).permutation(n)

    def consumed(self, executed_epochs):
        self.calls["train"] += 1

    def sample_noise(self, rows, A):
        k = self.calls["sample"]; self.calls["sample"] += 1
        return self._draw(self._rng("sample", k), (rows, A))

    def eval_noise(self, rows, A):
        k = self.calls["eval"]; self.calls["eval"] += 1
        return self._draw(self._rng("eval", k), (rows, A))


class RecordedStreams:
    """replays draws recorded from a run of the reference (tests/golden/g8_icrl_lgw.npz): discrete actions only."""

    def __init__(self, g):
        self.learn_actions = g["learn_actions"]            # [n_rollouts_total, T, N]
        self.perms = g["perms"]                            # [n_perms_total, T*N]
        self.sample_actions = g["sample_actions"]          # [n_iters, rows]
        self.eval_actions, self.eval_counts = g["eval_actions"], g["eval_counts"]
        self.i_roll = self.i_perm = self.i_sample = self.i_eval = self.eval_off = 0

    def rollout_noise(self, T, N, A):
        a = self.learn_actions[self.i_roll]; self.i_roll += 1
        assert a.shape == (T, N), (a.shape, T, N)
        return a.astype(np.float32)

    def permutation(self, epoch, n):
        k = self.i_perm + epoch
        # epochs the reference never executed (early stop / end of the recording) get a valid filler: never consumed
        if k >= len(self.perms) or self.perms[k].shape[0] != n:
            return np.arange(n)
        return self.perms[k]

    def consumed(self, executed_epochs):
        self.i_perm += int(executed_epochs)

    def sample_noise(self, rows, A):
        a = self.sample_actions[self.i_sample]; self.i_sample += 1
        assert a.shape[0] == rows, (a.shape, rows)
        return a.astype(np.float32)

    def eval_noise(self, rows, A):
        n = int(self.eval_counts[self.i_eval]); self.i_eval += 1
        out = np.zeros(rows, np.float32)                   # steps beyond the recording are never taken
        out[:n] = self.eval_actions[self.eval_off:self.eval_off + n]
        self.eval_off += n
        return out
