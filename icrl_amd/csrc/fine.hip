// Fine-grained entry points of the update for a host that keeps its MLPs in torch — gfx950.
//
// SURVEY.md section 8(b) lists them beside the fused `icrl_ppo_lag_train`: a maintainer of the reference who wants to leave
// `policy.evaluate_actions` and autograd where they are (ppo_lag.py:216-288) can replace, one at a time,
//   rollout_buffer.get(batch_size)            buffers.py:594-627      -> icrl_minibatch_gather   (env-major flat index -> [T, N] storage)
//   advantage normalisation                   ppo_lag.py:219-222      -> icrl_adv_stats          (mean, unbiased std; cost advantages centred only)
//   the loss terms and their gradients        ppo_lag.py:224-281      -> icrl_ppo_lag_loss_fwd_bwd (d loss / d log_prob, d v_r, d v_c, d entropy:
//                                                                        what `loss.backward()` would send into the networks' outputs)
//   clip_grad_norm_ + optimizer.step()        ppo_lag.py:283-288      -> icrl_clip_adam_step     (one flat buffer: torch's clip coefficient and
//                                                                        single-tensor Adam, bias corrections in double)
//   dual.update_parameter(average_cost)       dual_variable.py:47-57  -> icrl_dual_step          (nu stays on the device)
//   rollout_buffer.add(...)                   buffers.py:554-592      -> icrl_buffer_add         (one step's arrays -> row t)
//   compute_is_weights                        constraint_net.py:231-256 -> icrl_is_weights       (episode products, both KLs, weights)
//   the constraint-net loss + its gradients   constraint_net.py:188-202 -> icrl_cn_loss_fwd_bwd  (on the network's outputs)
// Same conventions as the rest of the library: caller-owned device buffers, a stream, hipError_t as int, no allocation, no sync.
// Plain kernels (a minibatch is 64..512 rows; these calls are launch-bound by construction — the fused persistent kernels are the
// fast path); every reduction is a fixed tree over one block, so results do not depend on the launch.
#include "common.h"

namespace icrl {
namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// out4: mean_r, 1 / (std_r + 1e-8) with torch's unbiased std, mean_c, std_r
__global__ void __launch_bounds__(256) adv_stats_kernel(const float* __restrict__ adv_r, const float* __restrict__ adv_c, int n, float* out4) {
  __shared__ float red[256];
  float sr = 0.f, sc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { sr += adv_r[i]; sc += adv_c[i]; }
  const float mean_r = block_sum_256(sr, red) / (float)n, mean_c = block_sum_256(sc, red) / (float)n;
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float d = adv_r[i] - mean_r; ss += d * d; }
  const float var = block_sum_256(ss, red) / (float)(n - 1);
  if (threadIdx.x == 0) { const float sd = sqrtf(var); out4[0] = mean_r; out4[1] = 1.f / (sd + 1e-8f); out4[2] = mean_c; out4[3] = sd; }
}

__global__ void __launch_bounds__(256) gather_kernel(icrl_buffer_t b, const int* __restrict__ flat_idx, int n, float* obs, float* act, float* old_lp,
                                                     float* adv_r, float* adv_c, float* ret_r, float* ret_c, float* old_v_r, float* old_v_c) {
  const int row = blockIdx.x;
  if (row >= n) return;
  const int i = flat_idx[row], env = i / b.T, t = i - env * b.T;      // buffers.py:53-65: flat index = env * T + t
  const size_t s = (size_t)t * b.N + env;
  for (int k = threadIdx.x; k < b.obs_dim; k += 256) obs[(size_t)row * b.obs_dim + k] = b.observations[s * b.obs_dim + k];
  if (act != nullptr)
    for (int k = threadIdx.x; k < b.act_store; k += 256) act[(size_t)row * b.act_store + k] = b.actions[s * b.act_store + k];
  if (threadIdx.x == 0) {
    if (old_lp) old_lp[row] = b.log_probs[s];
    if (adv_r) adv_r[row] = b.reward_advantages[s];
    if (adv_c) adv_c[row] = b.cost_advantages[s];
    if (ret_r) ret_r[row] = b.reward_returns[s];
    if (ret_c) ret_c[row] = b.cost_returns[s];
    if (old_v_r) old_v_r[row] = b.reward_values[s];
    if (old_v_c) old_v_c[row] = b.cost_values[s];
  }
}

// ppo_lag.py:219-281 on the networks' outputs of ONE minibatch (n <= 65536 rows, one block): terms[0..7] = loss, policy_loss,
// reward_value_loss, cost_value_loss, entropy_loss, approx_kl, clip_fraction, 0
__global__ void __launch_bounds__(256) loss_fwd_bwd_kernel(const float* __restrict__ lp, const float* __restrict__ old_lp, const float* __restrict__ adv_r,
                                                           const float* __restrict__ adv_c, const float* __restrict__ v_r, const float* __restrict__ v_c,
                                                           const float* __restrict__ ret_r, const float* __restrict__ ret_c, const float* __restrict__ old_v_r,
                                                           const float* __restrict__ old_v_c, const float* __restrict__ entropy, const float* __restrict__ nu_p,
                                                           icrl_ppo_hyper_t hp, int n, float* terms, float* d_lp, float* d_v_r, float* d_v_c, float* d_ent) {
  __shared__ float red[256];
  const float nu = nu_p[0], inv_n = 1.f / (float)n, clip = hp.clip_range;
  float sr = 0.f, sc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { sr += adv_r[i]; sc += adv_c[i]; }
  const float mean_r = block_sum_256(sr, red) * inv_n, mean_c = block_sum_256(sc, red) * inv_n;
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float d = adv_r[i] - mean_r; ss += d * d; }
  const float istd = 1.f / (sqrtf(block_sum_256(ss, red) / (float)(n - 1)) + 1e-8f);
  float q[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // sum min(s1, s2) | sum Ac ratio | clipped count | sum (old_lp - lp) | sum entropy | sum e_r^2 | sum e_c^2
  for (int i = threadIdx.x; i < n; i += 256) {
    const float ratio = __expf(lp[i] - old_lp[i]);
    const float Ar = (adv_r[i] - mean_r) * istd, Ac = adv_c[i] - mean_c;
    const float s1 = Ar * ratio, s2 = Ar * fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float gsel = (s1 <= s2) ? Ar : 0.f;                          // d min(s1, s2) / d ratio
    const float dlp = inv_n / (1.f + nu) * (-gsel + nu * Ac) * ratio;    // policy term
    // entropy term: with an analytic entropy the loss holds -mean(entropy); without one (entropy == NULL) the reference uses
    // -mean(-log_prob) (ppo_lag.py:258-262), whose gradient goes into log_prob
    float ent_i;
    if (entropy != nullptr) { ent_i = entropy[i]; d_lp[i] = dlp; if (d_ent) d_ent[i] = -hp.ent_coef * inv_n; }
    else { ent_i = -lp[i]; d_lp[i] = dlp + hp.ent_coef * inv_n; }
    auto value = [&](float v, float R, float old, float vclip, float coef, float* d_out, float& sq) {
      float vp = v, pass = 1.f;
      if (vclip >= 0.f) {
        const float dv = v - old;
        vp = old + fminf(fmaxf(dv, -vclip), vclip);
        pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
      }
      const float e = vp - R;
      d_out[i] = coef * 2.f * e * inv_n * pass;
      sq += e * e;
    };
    value(v_r[i], ret_r[i], old_v_r ? old_v_r[i] : 0.f, old_v_r ? hp.clip_range_reward_vf : -1.f, hp.reward_vf_coef, d_v_r, q[5]);
    value(v_c[i], ret_c[i], old_v_c ? old_v_c[i] : 0.f, old_v_c ? hp.clip_range_cost_vf : -1.f, hp.cost_vf_coef, d_v_c, q[6]);
    q[0] += fminf(s1, s2); q[1] += Ac * ratio; q[2] += fabsf(ratio - 1.f) > clip ? 1.f : 0.f; q[3] += old_lp[i] - lp[i]; q[4] += ent_i;
  }
  for (int k = 0; k < 7; ++k) q[k] = block_sum_256(q[k], red);
  if (threadIdx.x == 0) {
    const float policy_loss = (-(q[0] * inv_n) + nu * (q[1] * inv_n)) / (1.f + nu);
    const float rvl = q[5] * inv_n, cvl = q[6] * inv_n, entropy_loss = -(q[4] * inv_n);
    terms[0] = policy_loss + hp.ent_coef * entropy_loss + hp.reward_vf_coef * rvl + hp.cost_vf_coef * cvl;
    terms[1] = policy_loss; terms[2] = rvl; terms[3] = cvl; terms[4] = entropy_loss; terms[5] = q[3] * inv_n; terms[6] = q[2] * inv_n; terms[7] = 0.f;
  }
}

// clip_grad_norm_ (total = sqrt(sum g^2), coef = min(1, max_norm / (total + 1e-6))) and torch.optim.Adam, single-tensor form
__global__ void __launch_bounds__(256) sqnorm_partials_kernel(const float* __restrict__ g, long long n, float* part) {
  __shared__ float red[256];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s = fmaf(g[i], g[i], s);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) clip_adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, const int* adam_t, long long n,
                                                        icrl_ppo_hyper_t hp, const float* __restrict__ part, int n_part, float* out2) {
  __shared__ float red[256];
  __shared__ float bc[2];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
  const float total = sqrtf(block_sum_256(s, red));
  const float c = hp.max_grad_norm / (total + 1e-6f), coef = c > 1.f ? 1.f : c;
  if (threadIdx.x == 0) {
    const double t = (double)(adam_t[0] + 1);
    bc[0] = (float)((double)hp.lr / (1.0 - pow((double)hp.adam_beta1, t)));
    bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)hp.adam_beta2, t)));
    if (blockIdx.x == 0 && out2 != nullptr) { out2[0] = total; out2[1] = coef; }
  }
  __syncthreads();
  const float b1 = hp.adam_beta1, b2 = hp.adam_beta2, w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = fmaf(w1, gi, b1 * m[i]), vi = fmaf(w2, gi * gi, b2 * v[i]);
    m[i] = mi; v[i] = vi;
    p[i] = fmaf(-bc[0], mi / fmaf(sqrtf(vi), bc[1], hp.adam_eps), p[i]);
  }
}

// common/utils.py:43-59 explained_variance(y_pred, y_true) = 1 - Var[y_true - y_pred] / Var[y_true] (nan when Var[y_true] == 0) for up to two
// (y_pred, y_true) pairs of n floats in one pass: float64 sums of y, y^2, d, d^2 per block -> part[block][8], then one block folds them.
__device__ __forceinline__ double block_sum_256d(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) ev_partials_kernel(const float* __restrict__ pa, const float* __restrict__ ta, const float* __restrict__ pb,
                                                          const float* __restrict__ tb, long long n, double* part) {
  __shared__ double red[256];
  double q[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double ya = (double)ta[i], da = ya - (double)pa[i];
    q[0] += ya; q[1] += ya * ya; q[2] += da; q[3] += da * da;
    if (pb != nullptr) {
      const double yb = (double)tb[i], db = yb - (double)pb[i];
      q[4] += yb; q[5] += yb * yb; q[6] += db; q[7] += db * db;
    }
  }
  for (int k = 0; k < 8; ++k) {
    const double r = block_sum_256d(q[k], red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * 8 + k] = r;
  }
}

__global__ void __launch_bounds__(256) ev_final_kernel(const double* __restrict__ part, int n_part, long long n, int pairs, float* out) {
  __shared__ double red[256];
  __shared__ double tot[8];
  for (int k = 0; k < 8; ++k) {
    double s = 0.;
    for (int i = threadIdx.x; i < n_part; i += 256) s += part[(size_t)i * 8 + k];
    const double r = block_sum_256d(s, red);
    if (threadIdx.x == 0) tot[k] = r;
  }
  __syncthreads();
  if (threadIdx.x < pairs) {
    const double* t = tot + 4 * threadIdx.x;
    const double inv = 1.0 / (double)n, my = t[0] * inv, md = t[2] * inv;
    const double var_y = t[1] * inv - my * my, var_d = t[3] * inv - md * md;
    out[threadIdx.x] = var_y > 0.0 ? (float)(1.0 - var_d / var_y) : __builtin_nanf("");
  }
}

__global__ void bump_step_kernel(int* adam_t) { adam_t[0] += 1; }

// dual_variable.py:9-57 in float32: state = {log_nu, exp_avg, exp_avg_sq, nu (output)}, t = Adam step count
__global__ void dual_step_kernel(float* st, int* t_p, const float* cost_p, float cost_v, float alpha, float lr, float clamp_log_nu, float* loss_out) {
  const float cost = cost_p != nullptr ? cost_p[0] : cost_v;
  const float c = cost - alpha;
  const float x = st[0];
  const float z = expf(x);
  const float nu = x > 20.f ? x : log1pf(z);
  if (loss_out) loss_out[0] = -nu * c;
  const float sig = x > 20.f ? 1.f : z / (z + 1.f);
  const float g = (-c) * sig;
  const int t = t_p[0] + 1;
  t_p[0] = t;
  const float b1 = 0.9f, b2 = 0.999f;
  const float m = st[1] * b1 + (float)(1.0 - 0.9) * g;
  const float v = st[2] * b2 + (float)(1.0 - 0.999) * g * g;
  st[1] = m; st[2] = v;
  const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float denom = sqrtf(v) / (float)sqrt(bc2) + 1e-8f;
  float nx = x - step_size * (m / denom);
  nx = nx > clamp_log_nu ? nx : clamp_log_nu;
  st[0] = nx;
  st[3] = nx > 20.f ? nx : log1pf(expf(nx));
}


// RolloutBufferWithCost.add (buffers.py:554-592): one step's arrays of all envs -> row t of the [T, N] planes (float64 -> float32 like the
// reference's np.array(x).copy() into float32 storage).  One thread per (env, component).
__global__ void __launch_bounds__(256) buffer_add_kernel(icrl_buffer_t b, int t, const double* obs, const double* orig_obs, const double* new_obs,
                                                         const double* new_orig_obs, const float* action, const double* reward, const double* cost,
                                                         const float* orig_cost, const uint8_t* done, const float* reward_value, const float* cost_value,
                                                         const float* log_prob) {
  const int i = blockIdx.x * 256 + threadIdx.x, N = b.N, O = b.obs_dim, AS = b.act_store;
  const size_t row = (size_t)t * N;
  if (i < N * O) {
    b.observations[row * O + i] = (float)obs[i]; b.orig_observations[row * O + i] = (float)orig_obs[i];
    b.new_observations[row * O + i] = (float)new_obs[i]; b.new_orig_observations[row * O + i] = (float)new_orig_obs[i];
  }
  if (i < N * AS) b.actions[row * AS + i] = action[i];
  if (i < N) {
    b.rewards[row + i] = (float)reward[i]; b.costs[row + i] = (float)cost[i]; b.orig_costs[row + i] = orig_cost[i];
    b.dones[row + i] = (float)done[i]; b.reward_values[row + i] = reward_value[i]; b.cost_values[row + i] = cost_value[i];
    b.log_probs[row + i] = log_prob[i];
  }
}

// ConstraintNet.compute_is_weights (constraint_net.py:231-256) on the predictions of the start-of-call and of the current network:
// per-episode float32 products of (new + eps) / (old + eps) — they overflow to inf / nan for long episodes exactly like the reference's —,
// weights per step (ratio / mean ratio) or per episode (n_ep prod / (sum prod + eps), repeated over the episode's rows), both KLs.
// One workgroup; one wave per episode for the products (lane-strided, wave_prod: the association differs from torch.prod's by rounding).
__device__ __forceinline__ float wave_prod_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v *= __shfl_xor(v, m, 64);
  return v;
}
__global__ void __launch_bounds__(1024) is_weights_kernel(const float* __restrict__ old_p, const float* __restrict__ new_p, int N, const int* __restrict__ ep_off,
                                                          int n_ep, float eps, int per_step, float* w, float* ep_prod, float* out4) {
  __shared__ float red[16];
  __shared__ float sh[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int e = wv; e < n_ep; e += 16) {
    float pr = 1.f;
    for (int i = ep_off[e] + lane; i < ep_off[e + 1]; i += 64) pr *= (new_p[i] + eps) / (old_p[i] + eps);
    pr = wave_prod_f(pr);
    if (lane == 0) ep_prod[e] = pr;
  }
  float sr = 0.f;
  for (int i = tid; i < N; i += 1024) sr += (new_p[i] + eps) / (old_p[i] + eps);
  sr = wave_sum_fast(sr);
  if (lane == 0) red[wv] = sr;
  __syncthreads();
  if (tid == 0) {
    float sum_p = 0.f, tot = 0.f;
    for (int e = 0; e < n_ep; ++e) sum_p += ep_prod[e];
    for (int k = 0; k < 16; ++k) tot += red[k];
    const float pm = sum_p / (float)n_ep;
    float s1 = 0.f, s2 = 0.f;
    for (int e = 0; e < n_ep; ++e) {
      const float lp = logf(ep_prod[e] + eps);
      s1 += -lp;
      s2 += (ep_prod[e] - pm) * lp / (pm + eps);
    }
    out4[0] = s1 / (float)n_ep; out4[1] = s2 / (float)n_ep; out4[2] = tot / (float)N; out4[3] = sum_p;
    sh[0] = tot / (float)N; sh[1] = sum_p;
  }
  __syncthreads();
  if (per_step) {
    const float mr = sh[0];
    for (int i = tid; i < N; i += 1024) w[i] = ((new_p[i] + eps) / (old_p[i] + eps)) / mr;
  } else {
    for (int e = wv; e < n_ep; e += 16) {
      const float nw = (float)n_ep * ep_prod[e] / (sh[1] + eps);
      for (int i = ep_off[e] + lane; i < ep_off[e + 1]; i += 64) w[i] = nw;
    }
  }
}

// The constraint-net loss on the network's OUTPUTS (constraint_net.py:188-202) and d loss / d those outputs.  terms6 = {loss, expert_loss,
// nominal_loss, regularizer, mean(log(nominal + eps)), 0}.  mode bit 0: GAIL's BCE form; bit 1: the per-step broadcast quirk
// (weights [B,1,1] x log zeta [B,1] -> [B,B,1]: nominal_loss = mean(w) * mean(log zeta_N)).  w NULL: no importance sampling (ones).
__global__ void __launch_bounds__(256) cn_loss_fwd_bwd_kernel(const float* __restrict__ nom, const float* __restrict__ exp_, const float* __restrict__ w, int Bn, int Be,
                                                              float reg_coeff, float eps, int mode, float* terms, float* d_nom, float* d_exp) {
  __shared__ float red[256];
  const bool gail = mode & 1, factored = (mode & 2) && w != nullptr;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f;      // sum w log n | sum log n | sum w | sum (1 - n) | expert term | sum (1 - e)
  for (int i = threadIdx.x; i < Bn; i += 256) {
    const float n_ = nom[i], wi = w ? w[i] : 1.f;
    const float ln = gail ? -fmaxf(logf(1.f - n_), -100.f) : logf(n_ + eps);      // (torch BCELoss clamps its logs at -100)
    a0 += wi * ln; a1 += ln; a2 += wi; a3 += 1.f - n_;
  }
  for (int i = threadIdx.x; i < Be; i += 256) {
    const float e_ = exp_[i];
    a4 += gail ? -fmaxf(logf(e_), -100.f) : logf(e_ + eps);
    a5 += 1.f - e_;
  }
  a0 = block_sum_256(a0, red); a1 = block_sum_256(a1, red); a2 = block_sum_256(a2, red); a3 = block_sum_256(a3, red);
  a4 = block_sum_256(a4, red); a5 = block_sum_256(a5, red);
  const float inv_n = 1.f / (float)Bn, inv_e = 1.f / (float)Be, mean_w = a2 * inv_n;
  if (threadIdx.x == 0) {
    if (gail) {
      const float nl = a1 * inv_n, el = a4 * inv_e;
      terms[0] = nl + el; terms[1] = el; terms[2] = nl; terms[3] = 0.f; terms[4] = 0.f; terms[5] = 0.f;
    } else {
      const float el = a4 * inv_e, nl = factored ? mean_w * (a1 * inv_n) : a0 * inv_n, reg = reg_coeff * (a5 * inv_e + a3 * inv_n);
      terms[0] = (-el + nl) + reg; terms[1] = el; terms[2] = nl; terms[3] = reg; terms[4] = a1 * inv_n; terms[5] = 0.f;
    }
  }
  for (int i = threadIdx.x; i < Bn; i += 256) {
    const float n_ = nom[i], wi = factored ? mean_w : (w ? w[i] : 1.f);
    d_nom[i] = gail ? (logf(1.f - n_) > -100.f ? inv_n / (1.f - n_) : 0.f) : wi * inv_n / (n_ + eps) - reg_coeff * inv_n;
  }
  for (int i = threadIdx.x; i < Be; i += 256) {
    const float e_ = exp_[i];
    d_exp[i] = gail ? (logf(e_) > -100.f ? -inv_e / e_ : 0.f) : -inv_e / (e_ + eps) - reg_coeff * inv_e;
  }
}

}  // namespace
}  // namespace icrl

using namespace icrl;

extern "C" int icrl_adv_stats(const float* adv_r, const float* adv_c, int n, float* out4, void* stream) {
  if (n < 2 || adv_r == nullptr || adv_c == nullptr || out4 == nullptr) return fail("icrl_adv_stats: n = %d (>= 2), NULL argument", n);
  hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, adv_r, adv_c, n, out4);
  return (int)hipGetLastError();
}

extern "C" int icrl_minibatch_gather(const icrl_buffer_t* buf, const int32_t* flat_idx, int n, float* obs, float* actions, float* old_log_prob,
                                     float* adv_r, float* adv_c, float* ret_r, float* ret_c, float* old_v_r, float* old_v_c, void* stream) {
  if (buf == nullptr || flat_idx == nullptr || obs == nullptr || n < 1) return fail("icrl_minibatch_gather: n = %d, NULL argument", n);
  hipLaunchKernelGGL(gather_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, *buf, flat_idx, n, obs, actions, old_log_prob, adv_r, adv_c, ret_r, ret_c,
                     old_v_r, old_v_c);
  return (int)hipGetLastError();
}

extern "C" int icrl_ppo_lag_loss_fwd_bwd(const float* log_prob, const float* old_log_prob, const float* adv_r, const float* adv_c, const float* v_r,
                                         const float* v_c, const float* ret_r, const float* ret_c, const float* old_v_r, const float* old_v_c,
                                         const float* entropy, const float* nu, const icrl_ppo_hyper_t* hp, int n, float* terms8, float* d_log_prob,
                                         float* d_v_r, float* d_v_c, float* d_entropy, void* stream) {
  if (n < 2 || n > 65536) return fail("icrl_ppo_lag_loss_fwd_bwd: n = %d rows (2..65536: one minibatch)", n);
  if (!log_prob || !old_log_prob || !adv_r || !adv_c || !v_r || !v_c || !ret_r || !ret_c || !nu || !hp || !terms8 || !d_log_prob || !d_v_r || !d_v_c)
    return fail("icrl_ppo_lag_loss_fwd_bwd: NULL argument (only old_v_r / old_v_c — no value clipping —, entropy and d_entropy may be NULL)");
  hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, log_prob, old_log_prob, adv_r, adv_c, v_r, v_c, ret_r, ret_c, old_v_r,
                     old_v_c, entropy, nu, *hp, n, terms8, d_log_prob, d_v_r, d_v_c, d_entropy);
  return (int)hipGetLastError();
}

extern "C" int icrl_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, long long n,
                                   const icrl_ppo_hyper_t* hp, float* work, float* out2, void* stream) {
  if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !adam_step || !hp || !work) return fail("icrl_clip_adam_step: n = %lld, NULL argument (work: 256 floats)", n);
  const int blocks = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(blocks), dim3(256), 0, s, grads, n, work);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, adam_step, n, *hp, work, blocks, out2);
  hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, s, adam_step);
  return (int)hipGetLastError();
}

extern "C" int icrl_explained_variance(const float* y_pred_a, const float* y_true_a, const float* y_pred_b, const float* y_true_b, long long n,
                                       double* work, float* out2, void* stream) {
  if (n < 1 || !y_pred_a || !y_true_a || !work || !out2 || ((y_pred_b == nullptr) != (y_true_b == nullptr)))
    return fail("icrl_explained_variance: n = %lld, NULL argument (the second pair may be NULL as a whole; work: 8 x 256 doubles)", n);
  const int blocks = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ev_partials_kernel, dim3(blocks), dim3(256), 0, s, y_pred_a, y_true_a, y_pred_b, y_true_b, n, work);
  hipLaunchKernelGGL(ev_final_kernel, dim3(1), dim3(256), 0, s, work, blocks, n, y_pred_b != nullptr ? 2 : 1, out2);
  return (int)hipGetLastError();
}

extern "C" int icrl_dual_step(float* state4, int32_t* adam_step, const float* cost_dev, float cost_host, float budget, float learning_rate,
                              float clamp_log_nu, float* loss_out, void* stream) {
  if (!state4 || !adam_step) return fail("icrl_dual_step: NULL state");
  hipLaunchKernelGGL(dual_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state4, adam_step, cost_dev, cost_host, budget, learning_rate, clamp_log_nu, loss_out);
  return (int)hipGetLastError();
}

extern "C" int icrl_buffer_add(const icrl_buffer_t* buf, int t, const double* obs, const double* orig_obs, const double* new_obs, const double* new_orig_obs,
                               const float* action, const double* reward, const double* cost, const float* orig_cost, const uint8_t* done,
                               const float* reward_value, const float* cost_value, const float* log_prob, void* stream) {
  if (buf == nullptr || t < 0 || t >= buf->T) return fail("icrl_buffer_add: t = %d outside the buffer's %d steps", t, buf ? buf->T : 0);
  if (!obs || !orig_obs || !new_obs || !new_orig_obs || !action || !reward || !cost || !orig_cost || !done || !reward_value || !cost_value || !log_prob)
    return fail("icrl_buffer_add: NULL argument");
  const int m = buf->N * (buf->obs_dim > buf->act_store ? buf->obs_dim : buf->act_store);
  hipLaunchKernelGGL(buffer_add_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, *buf, t, obs, orig_obs, new_obs, new_orig_obs, action, reward, cost,
                     orig_cost, done, reward_value, cost_value, log_prob);
  return (int)hipGetLastError();
}

extern "C" int icrl_is_weights(const float* preds_old, const float* preds_new, int N, const int32_t* ep_offsets, int n_ep, float eps, int per_step,
                               float* weights, float* ep_prod, float* out4, void* stream) {
  if (N < 1 || n_ep < 1 || !preds_old || !preds_new || !ep_offsets || !weights || !ep_prod || !out4) return fail("icrl_is_weights: N = %d, n_ep = %d, NULL argument", N, n_ep);
  hipLaunchKernelGGL(is_weights_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, preds_old, preds_new, N, ep_offsets, n_ep, eps, per_step, weights, ep_prod, out4);
  return (int)hipGetLastError();
}

extern "C" int icrl_cn_loss_fwd_bwd(const float* nominal_preds, const float* expert_preds, const float* is_weights, int Bn, int Be, float reg_coeff, float eps,
                                    int mode, float* terms6, float* d_nominal, float* d_expert, void* stream) {
  if (Bn < 1 || Be < 1 || !nominal_preds || !expert_preds || !terms6 || !d_nominal || !d_expert) return fail("icrl_cn_loss_fwd_bwd: Bn = %d, Be = %d, NULL argument", Bn, Be);
  hipLaunchKernelGGL(cn_loss_fwd_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nominal_preds, expert_preds, is_weights, Bn, Be, reg_coeff, eps, mode, terms6,
                     d_nominal, d_expert);
  return (int)hipGetLastError();
}
