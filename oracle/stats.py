"""Oracle: running moments and the obs / reward / cost normaliser.  Test infrastructure only.

ref: stable_baselines3/common/running_mean_std.py:6-39          (RunningMeanStd)
     stable_baselines3/common/vec_env/vec_normalize.py:81-123    (VecNormalize.step_wait & co)
     stable_baselines3/common/vec_env/vec_normalize.py:184-278   (VecNormalizeWithCost)

Everything here is float64, as in the reference (numpy defaults).  The functions are
written array-in / array-out over a small state object instead of as env wrappers:
the wrapper classes that mirror the reference's API live in ``icrl_amd`` (product) and
in ``oracle/loop.py`` (CPU port); both call the same arithmetic in the same order.
"""
import numpy as np

F64 = np.float64


class Moments:
    """mean / var / count triple merged with Chan's parallel formula.
    ref: running_mean_std.py:6-39 (count starts at epsilon = 1e-4, var at 1)."""

    def __init__(self, shape=(), epsilon=1e-4):
        self.mean = np.zeros(shape, F64)
        self.var = np.ones(shape, F64)
        self.count = float(epsilon)

    def copy(self):
        m = Moments(self.mean.shape)
        m.mean, m.var, m.count = self.mean.copy(), self.var.copy(), self.count
        return m

    def update(self, batch):
        """ref: running_mean_std.py:19-23 — biased batch variance (np.var)."""
        batch = np.asarray(batch, dtype=F64)
        self.merge(np.mean(batch, axis=0), np.var(batch, axis=0), batch.shape[0])

    def merge(self, b_mean, b_var, b_count):
        """ref: running_mean_std.py:25-39, operation order preserved."""
        delta = b_mean - self.mean
        tot = self.count + b_count
        new_mean = self.mean + delta * b_count / tot
        m_a = self.var * self.count
        m_b = b_var * b_count
        m_2 = m_a + m_b + np.square(delta) * self.count * b_count / (self.count + b_count)
        self.mean = new_mean
        self.var = m_2 / (self.count + b_count)
        self.count = b_count + self.count


class NormState:
    """State of VecNormalizeWithCost (ref: vec_normalize.py:23-37,184-199)."""

    def __init__(self, n_envs, obs_dim, training=True, norm_obs=True, norm_reward=True,
                 norm_cost=True, clip_obs=10.0, clip_reward=10.0, clip_cost=10.0,
                 reward_gamma=0.99, cost_gamma=0.99, epsilon=1e-8):
        self.obs_rms = Moments((obs_dim,))
        self.ret_rms = Moments(())
        self.cost_rms = Moments(())
        self.ret = np.zeros(n_envs, F64)
        self.cost_ret = np.zeros(n_envs, F64)
        self.training, self.norm_obs, self.norm_reward, self.norm_cost = training, norm_obs, norm_reward, norm_cost
        self.clip_obs, self.clip_reward, self.clip_cost = clip_obs, clip_reward, clip_cost
        self.reward_gamma, self.cost_gamma, self.epsilon = reward_gamma, cost_gamma, epsilon


def normalize_obs(st, obs):
    """ref: vec_normalize.py:107-114."""
    if not st.norm_obs:
        return obs
    return np.clip((obs - st.obs_rms.mean) / np.sqrt(st.obs_rms.var + st.epsilon), -st.clip_obs, st.clip_obs)


def normalize_scalar(rms, x, clip, eps, enabled):
    """ref: vec_normalize.py:116-123 (reward) / :254-261 (cost): no mean subtraction."""
    if not enabled:
        return x
    return np.clip(x / np.sqrt(rms.var + eps), -clip, clip)


def norm_reset(st, raw_obs):
    """ref: vec_normalize.py:148-157 + :270-278.  Zeroes the discounted returns and feeds a
    zero batch into ret_rms / cost_rms (when training); obs_rms is NOT updated on reset."""
    n = st.ret.shape[0]
    st.ret = np.zeros(n, F64)
    if st.training:
        st.ret = st.ret * st.reward_gamma + st.ret
        st.ret_rms.update(st.ret)
    out = normalize_obs(st, np.asarray(raw_obs, F64))
    st.cost_ret = np.zeros(n, F64)
    if st.training:
        st.cost_ret = st.cost_ret * st.cost_gamma + st.cost_ret
        st.cost_rms.update(st.cost_ret)
    return out


def norm_step(st, raw_obs, raw_rew, raw_cost, dones):
    """One wrapper step.  ref: vec_normalize.py:81-100 then :220-243.
    raw_obs [N, obs] f64, raw_rew [N] f64, raw_cost [N] (f32 from the cost net) or None,
    dones [N] bool.  Returns (obs_n, rew_n, cost_n)."""
    raw_obs = np.asarray(raw_obs, F64)
    raw_rew = np.asarray(raw_rew, F64)
    if st.training:
        st.obs_rms.update(raw_obs)                       # stats first, then normalise
    obs_n = normalize_obs(st, raw_obs)
    if st.training:
        st.ret = st.ret * st.reward_gamma + raw_rew
        st.ret_rms.update(st.ret)
    rew_n = normalize_scalar(st.ret_rms, raw_rew, st.clip_reward, st.epsilon, st.norm_reward)
    st.ret[dones] = 0
    cost_n = None
    if raw_cost is not None:
        raw_cost = np.asarray(raw_cost)
        if st.training:
            st.cost_ret = st.cost_ret * st.cost_gamma + raw_cost
            st.cost_rms.update(st.cost_ret)
        cost_n = normalize_scalar(st.cost_rms, raw_cost, st.clip_cost, st.epsilon, st.norm_cost)
        st.cost_ret[dones] = 0
    return obs_n, rew_n, cost_n
