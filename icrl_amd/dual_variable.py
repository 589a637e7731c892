"""Lagrange multiplier nu = softplus(log_nu) with a scalar Adam step per PPO iteration.

ref: stable_baselines3/common/dual_variable.py:9-57 (Nu, DualVariable), :60-122 (PIDLagrangian).
A single float32 scalar + Adam state: there is nothing to parallelise, so it is host arithmetic in numpy float32 with
the same operation order as torch's softplus / Adam (checked against the reference's golden trajectories, tests/golden/g5).
"""
from collections import deque

import numpy as np

F32 = np.float32


def _inv_softplus_floor(x):
    return np.log(max(np.exp(x) - 1, 1e-8))


class _Scalar:
    """quacks like the 0-d tensor the reference returns from nu() / loss (``.item()``)."""

    def __init__(self, v):
        self.v = v

    def item(self):
        return float(self.v)

    def __float__(self):
        return float(self.v)


class DualVariable:
    def __init__(self, alpha=0, learning_rate=10, penalty_init=1, clamp_at=None):
        init = _inv_softplus_floor(penalty_init)
        self.penalty_init = penalty_init
        self.log_nu = F32(init)
        # quirk kept: clamp_at defaults to the already inverse-softplused init and clamp() inverts it again
        self.clamp_at = init if clamp_at is None else clamp_at
        self.alpha, self.lr = alpha, learning_rate
        self.m, self.v, self.t = F32(0), F32(0), 0
        self.loss = _Scalar(0.0)

    @staticmethod
    def _softplus(x):
        # torch softplus (beta=1, threshold=20): x if x > 20 else log1p(exp(x)), float32
        x = F32(x)
        return x if x > F32(20) else F32(np.log1p(np.exp(x, dtype=F32), dtype=F32))

    def nu(self):
        return _Scalar(self._softplus(self.log_nu))

    def state_dict(self):
        return dict(log_nu=float(self.log_nu), adam_m=float(self.m), adam_v=float(self.v), adam_t=int(self.t))

    def load_state_dict(self, sd):
        self.log_nu, self.m, self.v, self.t = F32(sd["log_nu"]), F32(sd["adam_m"]), F32(sd["adam_v"]), int(sd["adam_t"])

    def update_parameter(self, cost):
        """ref: dual_variable.py:47-57: loss = -nu * (cost - alpha); Adam(lr, betas (0.9, 0.999), eps 1e-8); clamp."""
        c = F32(cost - self.alpha)
        nu = self._softplus(self.log_nu)
        self.loss = _Scalar(F32(-nu * c))
        # d loss / d log_nu = -c * sigmoid(log_nu)   (softplus' = sigmoid; torch: z = exp(x); z / (z + 1))
        z = np.exp(self.log_nu, dtype=F32)
        sig = F32(1) if self.log_nu > F32(20) else F32(z / (z + F32(1)))
        g = F32(F32(-c) * sig)
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m = F32(self.m * F32(b1) + F32(1 - b1) * g)      # lerp(m, g, 1-b1) == m + (g - m) * (1 - b1) in torch >= 2
        self.v = F32(self.v * F32(b2) + F32(1 - b2) * g * g)
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        step_size = self.lr / bc1
        denom = F32(F32(np.sqrt(self.v, dtype=F32)) / F32(np.sqrt(bc2)) + F32(eps))
        self.log_nu = F32(self.log_nu - F32(step_size) * F32(self.m / denom))
        self.log_nu = max(self.log_nu, F32(_inv_softplus_floor(self.clamp_at)))


class PIDLagrangian:
    """ref: dual_variable.py:60-122 (used by cpg --use_pid only)."""

    def __init__(self, alpha=0, penalty_init=1, Kp=0, Kd=0, Ki=1, pid_delay=10, delta_d_ema_alpha=0.95, delta_p_ema_alpha=0.95):
        self.budget, self.Kp, self.Ki, self.Kd, self.pid_delay = alpha, Kp, Ki, Kd, pid_delay
        self.pid_i = self.cost_penalty = penalty_init
        self.cost_deltas = deque(maxlen=pid_delay)
        self.cost_deltas.append(0)
        self._delta_p = self._cost_delta = 0
        self.delta_d_ema_alpha, self.delta_p_ema_alpha = delta_d_ema_alpha, delta_p_ema_alpha
        self.loss = _Scalar(0.0)

    def update_parameter(self, cost):
        cost = float(cost)
        self.loss = _Scalar(F32(cost))        # torch.tensor(cost): float32, kept for interface parity with DualVariable
        delta = cost - self.budget
        self.pid_i = max(0, self.pid_i + self.Ki * delta)
        self._delta_p = self.delta_p_ema_alpha * self._delta_p + (1 - self.delta_p_ema_alpha) * delta
        self._cost_delta = self.delta_d_ema_alpha * self._cost_delta + (1 - self.delta_d_ema_alpha) * cost
        pid_d = max(0, self._cost_delta - self.cost_deltas[0])
        self.cost_penalty = max(0, self.Kp * self._delta_p + self.Kd * pid_d + self.pid_i)
        self.cost_deltas.append(self._cost_delta)

    def nu(self):
        return _Scalar(F32(self.cost_penalty))      # the reference hands it out as torch.tensor(float): float32
