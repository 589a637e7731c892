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

extern "C" int icrl_dual_step(float* state4, int32_t* adam_step, const float* cost_dev, float cost_host, float budget, float learning_rate,
                              float clamp_log_nu, float* loss_out, void* stream) {
  if (!state4 || !adam_step) return fail("icrl_dual_step: NULL state");
  hipLaunchKernelGGL(dual_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state4, adam_step, cost_dev, cost_host, budget, learning_rate, clamp_log_nu, loss_out);
  return (int)hipGetLastError();
}
