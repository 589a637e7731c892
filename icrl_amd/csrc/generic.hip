// Generic-shape path of the policy kernels — gfx950.
//
// The fast kernels (rollout.hip, ppo_train_*.hip) are built around 64-wide hidden layers, minibatches of at most 256 rows and one
// lane per hidden unit.  The reference accepts any `-pl / -rvl / -cvl` widths and any batch size (icrl/utils.py:636-655,
// stable_baselines3/common/torch_layers.py:129-254, common/buffers.py:594-612); this file serves what the fast kernels refuse:
// two-layer branches up to 256 units (stored padded to a common width HP = a multiple of 64; pad units have zero weights and
// biases, output tanh(0) = 0 and receive zero gradients, exactly like the narrow widths of the fast kernels) and minibatches of
// any size.  Plain kernels, one thread per hidden unit / per parameter, sequential fmaf chains, several launches per optimiser
// step — correct and deterministic, not latency-tuned: every BASELINE config runs on the fast kernels.
//
//   icrl_policy_forward / icrl_policy_evaluate      -> policy_generic_kernel            (policies.py:716-731, 752-767)
//   icrl_ppo_lag_train                              -> per optimiser step: gen_stats | gen_forward_backward | gen_wgrad | gen_adam
//                                                      (ppo_lag.py:196-299, torch.optim.Adam, clip_grad_norm_)
// Only `params` is read (never the transposed copy: the update changes the weights between its own launches).
#include "ppo_common.h"

namespace icrl {

constexpr int GEN_MAX_H = 256;

struct GenCtl {      // first 64 floats of the generic scratch, zeroed at the start of a train() launch sequence
  float mean_r, istd_r, mean_c;
  int stop, steps_done, early_stop_epoch;
  float kl_acc;
  float acc_ent, acc_pg, acc_cf, acc_vl_r, acc_vl_c, last_pol, last_vl_r, last_vl_c, mean_kl;
};

struct GenArgs {
  PolLayout L;
  int HP, B;                 // padded hidden width (= L.H1 = L.H2), batch_size
  float* params; float* exp_avg; float* exp_avg_sq;
  const int* adam_t;
  icrl_buffer_t buf;
  const int* perm_off;       // [n_epochs * T*N] storage offsets (ppo_perm_offsets_kernel)
  const float* nu;
  icrl_ppo_hyper_t hp;
  float* stats;
  float* scratch;            // ICRL_PPO_GENERIC_BYTES
  int n_total, n_mb;
};

// scratch layout in floats
__host__ __device__ inline size_t gen_off_rowstat() { return 64; }
__host__ __device__ inline size_t gen_off_rowidx(int B) { return gen_off_rowstat() + (size_t)3 * B * 8; }
__host__ __device__ inline size_t gen_off_g2(int B) { return gen_off_rowidx(B) + (size_t)B; }
__host__ __device__ inline size_t gen_off_act(int B, int role, int HP) { return gen_off_g2(B) + (size_t)B * 16 + (size_t)role * B * (4 * HP + 16); }
__host__ __device__ inline size_t gen_off_grad(int B, int HP) { return gen_off_act(B, 3, HP); }
__host__ __device__ inline size_t gen_off_part(int B, int HP, int n_params) { return gen_off_grad(B, HP) + (size_t)n_params; }
__host__ __device__ inline size_t gen_floats(int B, int HP, int n_params) { return gen_off_part(B, HP, n_params) + (size_t)(n_params + 255) / 256 + 64; }

// ---------------------------------------------------------------------------------------------------------------
// forward of ONE row through the three branches: thread (role, j) = hidden unit j of branch role.  blockDim = 3 * HP.
// ---------------------------------------------------------------------------------------------------------------
struct GenFwdShared {
  float x[1024];
  float h1[3][GEN_MAX_H], h2[3][GEN_MAX_H];
  float out[MAX_ACT];
  float scal[4];
};

__device__ __forceinline__ void gen_mlp_forward(const PolLayout& L, const float* __restrict__ P, GenFwdShared& sh, int role, int j) {
  const int O = L.O, H1 = L.H1, H2 = L.H2;
  {
    const float* w = P + L.W1[role] + (size_t)j * O;
    float z = P[L.b1[role] + j];
    for (int k = 0; k < O; ++k) z = fmaf(w[k], sh.x[k], z);
    sh.h1[role][j] = fast_tanh(z);
  }
  __syncthreads();
  {
    const float* w = P + L.W2[role] + (size_t)j * H1;
    float z = P[L.b2[role] + j];
    for (int k = 0; k < H1; ++k) z = fmaf(w[k], sh.h1[role][k], z);
    sh.h2[role][j] = fast_tanh(z);
  }
  __syncthreads();
  const int n_out = role == 0 ? L.A : 1;
  if (j < n_out) {
    const int Wh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc), bh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);
    const float* w = P + Wh + (size_t)j * H2;
    float z = P[bh + j];
    for (int k = 0; k < H2; ++k) z = fmaf(w[k], sh.h2[role][k], z);
    if (role == 0) sh.out[j] = z; else sh.scal[role - 1] = z;
  }
  __syncthreads();
}

// policies.py:716-731 (forward: sample / deterministic, clip, log-prob) and :752-767 (evaluate_actions: `given` actions, entropy)
__global__ void __launch_bounds__(3 * GEN_MAX_H) policy_generic_kernel(PolLayout L, const float* __restrict__ P, const double* __restrict__ obs,
                                                                        const float* __restrict__ noise, int deterministic, const float* alow,
                                                                        const float* ahigh, float* actions, float* act_clipped, float* v_r,
                                                                        float* v_c, float* log_prob, const float* __restrict__ given,
                                                                        float* entropy) {
  __shared__ GenFwdShared sh;
  const int HP = L.H1, tid = threadIdx.x, role = tid / HP, j = tid - role * HP;
  const size_t n = blockIdx.x;
  for (int i = tid; i < L.O; i += blockDim.x) sh.x[i] = (float)obs[n * L.O + i];
  __syncthreads();
  gen_mlp_forward(L, P, sh, role, j);
  if (tid == 0) {
    const int A = L.A, AS = L.discrete ? 1 : A;
    float lp = 0.f, ent = 0.f;
    if (!L.discrete) {
      for (int o = 0; o < A; ++o) {
        const float ls = P[L.log_std + o], sd = __expf(ls), mean = sh.out[o];
        float act = mean;
        if (given != nullptr) act = given[n * AS + o];
        else if (!deterministic && noise != nullptr) act = mean + noise[n * AS + o] * sd;      // Normal.rsample: loc + eps * scale
        const float diff = act - mean;
        lp += -(diff * diff) / (2.f * sd * sd) - ls - LOG_SQRT_2PI_F;
        ent += HALF_LOG_2PI_PLUS_HALF_F + ls;
        if (actions) actions[n * AS + o] = act;
        if (act_clipped) act_clipped[n * AS + o] = (alow != nullptr && ahigh != nullptr) ? fminf(fmaxf(act, alow[o]), ahigh[o]) : act;
      }
    } else {      // Categorical(logits): log-softmax, inverse-CDF sample on the injected uniform (spec: oracle/nets.py forward)
      float m = -INFINITY;
      for (int o = 0; o < A; ++o) m = fmaxf(m, sh.out[o]);
      float se = 0.f;
      for (int o = 0; o < A; ++o) se += expf(sh.out[o] - m);
      const float lse = m + logf(se);
      int action = 0;
      if (given != nullptr) action = (int)given[n];
      else if (deterministic || noise == nullptr) {
        float best = -1.f;
        for (int o = 0; o < A; ++o) { const float p = expf(sh.out[o] - lse); if (p > best) { best = p; action = o; } }
      } else {
        const float u = noise[n];
        float cdf = 0.f;
        int cnt = 0;
        for (int o = 0; o < A; ++o) { cdf += expf(sh.out[o] - lse); cnt += (u >= cdf) ? 1 : 0; }
        action = cnt < A - 1 ? cnt : A - 1;
      }
      lp = sh.out[action] - lse;
      for (int o = 0; o < A; ++o) { const float lg = sh.out[o] - lse; ent -= lg * expf(lg); }
      if (actions) actions[n] = (float)action;
      if (act_clipped) act_clipped[n] = (float)action;
    }
    if (v_r) v_r[n] = sh.scal[0];
    if (v_c) v_c[n] = sh.scal[1];
    if (log_prob) log_prob[n] = lp;
    if (entropy) entropy[n] = ent;
  }
}

int launch_policy_generic(const icrl_policy_t* p, const double* obs, const float* noise, int N, int deterministic, const float* alow,
                          const float* ahigh, float* actions, float* act_clipped, float* v_r, float* v_c, float* log_prob,
                          const float* given, float* entropy, hipStream_t s) {
  if (p->h1 != p->h2 || p->h1 % 64 != 0 || p->h1 > GEN_MAX_H || p->obs_dim < 1 || p->obs_dim > 1024 || p->act_dim < 1 || p->act_dim > MAX_ACT)
    return fail("policy forward / evaluate (generic path): obs_dim %d (1..1024), act_dim %d (1..%d), padded hidden width %d x %d (equal, a multiple "
                "of 64, <= %d)", p->obs_dim, p->act_dim, MAX_ACT, p->h1, p->h2, GEN_MAX_H);
  PolLayout L = make_pol_layout(p->obs_dim, p->act_dim, p->h1, p->h2, p->discrete);
  hipLaunchKernelGGL(policy_generic_kernel, dim3(N), dim3(3 * p->h1), 0, s, L, p->params, obs, noise, deterministic, alow, ahigh, actions,
                     act_clipped, v_r, v_c, log_prob, given, entropy);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// the update, one optimiser step = four launches; `step` (0-based index in the launch sequence) comes from the host
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {      // fixed tree over the block: deterministic
  const int tid = threadIdx.x, nt = blockDim.x;
  red[tid] = v;
  __syncthreads();
  for (int s = nt >> 1; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// advantage statistics of the minibatch (ppo_lag.py:219-222: reward advantages standardised with torch's unbiased std, cost
// advantages centred only)
__global__ void __launch_bounds__(256) gen_stats_kernel(GenArgs a, int perm_base, int nb) {
  __shared__ float red[256];
  GenCtl* ctl = reinterpret_cast<GenCtl*>(a.scratch);
  if (ctl->stop) return;
  float sr = 0.f, sc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { const int idx = a.perm_off[perm_base + i]; sr += a.buf.reward_advantages[idx]; sc += a.buf.cost_advantages[idx]; }
  const float mean_r = block_sum(sr, red) / (float)nb, mean_c = block_sum(sc, red) / (float)nb;
  float ss = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { const float d = a.buf.reward_advantages[a.perm_off[perm_base + i]] - mean_r; ss += d * d; }
  const float var = block_sum(ss, red) / (float)(nb - 1);
  if (threadIdx.x == 0) { ctl->mean_r = mean_r; ctl->mean_c = mean_c; ctl->istd_r = 1.f / (sqrtf(var) + 1e-8f); }
}

// forward, loss and activation backward of ONE minibatch row through ONE branch: grid (nb, 3), block HP
__global__ void __launch_bounds__(GEN_MAX_H) gen_forward_backward_kernel(GenArgs a, int perm_base, int nb) {
  __shared__ GenFwdShared sh;
  __shared__ float dout[MAX_ACT];
  const GenCtl* ctl = reinterpret_cast<const GenCtl*>(a.scratch);
  if (ctl->stop) return;
  const PolLayout& L = a.L;
  const int HP = a.HP, B = a.B, j = threadIdx.x, row = blockIdx.x, role = blockIdx.y;
  const float* P = a.params;
  const int idx = a.perm_off[perm_base + row];
  for (int i = j; i < L.O; i += HP) sh.x[i] = a.buf.observations[(size_t)idx * L.O + i];
  __syncthreads();
  gen_mlp_forward(L, P, sh, role, j);
  float* act = a.scratch + gen_off_act(B, role, HP);
  float* H1b = act, *H2b = act + (size_t)B * HP, *DZ1 = act + (size_t)2 * B * HP, *DZ2 = act + (size_t)3 * B * HP, *DOUT = act + (size_t)4 * B * HP;
  H1b[(size_t)row * HP + j] = sh.h1[role][j];
  H2b[(size_t)row * HP + j] = sh.h2[role][j];
  const int n_out = role == 0 ? L.A : 1;
  if (j == 0) {
    const float nu = a.nu[0], inv_nb = 1.f / (float)nb;
    float* rs = a.scratch + gen_off_rowstat() + ((size_t)role * B + row) * 8;
    for (int o = 0; o < MAX_ACT; ++o) dout[o] = 0.f;
    if (role == 0) {
      reinterpret_cast<int*>(a.scratch + gen_off_rowidx(B))[row] = idx;
      float* g2row = a.scratch + gen_off_g2(B) + (size_t)row * 16;
      const int A = L.A;
      float lp = 0.f, ent = 0.f, g1[MAX_ACT], g2[MAX_ACT];
      if (!L.discrete) {
        for (int o = 0; o < A; ++o) {
          const float ls = P[L.log_std + o], sd = __expf(ls), iv = 1.f / (sd * sd);
          const float dd = a.buf.actions[(size_t)idx * a.buf.act_store + o] - sh.out[o];
          lp += -(dd * dd) * (0.5f * iv) - ls - LOG_SQRT_2PI_F;
          g1[o] = dd * iv;
          g2[o] = (dd * dd) * iv - 1.f;
          ent += HALF_LOG_2PI_PLUS_HALF_F + ls;
        }
      } else {
        float m = -INFINITY;
        for (int o = 0; o < A; ++o) m = fmaxf(m, sh.out[o]);
        float se = 0.f;
        for (int o = 0; o < A; ++o) se += expf(sh.out[o] - m);
        const float lse = m + logf(se);
        const int action = (int)a.buf.actions[(size_t)idx * a.buf.act_store];
        for (int o = 0; o < A; ++o) { const float lg = sh.out[o] - lse; ent -= expf(lg) * lg; }
        for (int o = 0; o < A; ++o) {
          const float lg = sh.out[o] - lse, pr = expf(lg);
          if (o == action) lp = lg;
          g1[o] = (o == action ? 1.f : 0.f) - pr;
          g2[o] = pr * (lg + ent);
        }
      }
      const float old_lp = a.buf.log_probs[idx];
      const float ratio = __expf(lp - old_lp);
      const float Ar = (a.buf.reward_advantages[idx] - ctl->mean_r) * ctl->istd_r;
      const float Ac = a.buf.cost_advantages[idx] - ctl->mean_c;
      const float clip = a.hp.clip_range;
      const float s1 = Ar * ratio, s2 = Ar * fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
      const float gsel = (s1 <= s2) ? Ar : 0.f;
      const float dlp = inv_nb / (1.f + nu) * (-gsel + nu * Ac) * ratio;
      for (int o = 0; o < A; ++o) {
        dout[o] = L.discrete ? dlp * g1[o] + a.hp.ent_coef * inv_nb * g2[o] : dlp * g1[o];
        g2row[o] = L.discrete ? 0.f : dlp * g2[o];
      }
      rs[0] = fminf(s1, s2); rs[1] = Ac * ratio; rs[2] = fabsf(ratio - 1.f) > clip ? 1.f : 0.f; rs[3] = old_lp - lp; rs[4] = ent;
    } else {
      const float v = sh.scal[role - 1];
      const float R = role == 1 ? a.buf.reward_returns[idx] : a.buf.cost_returns[idx];
      const float vclip = role == 1 ? a.hp.clip_range_reward_vf : a.hp.clip_range_cost_vf;
      const float vcoef = role == 1 ? a.hp.reward_vf_coef : a.hp.cost_vf_coef;
      float vp = v, pass = 1.f;
      if (vclip >= 0.f) {
        const float old = role == 1 ? a.buf.reward_values[idx] : a.buf.cost_values[idx];
        const float dv = v - old;
        vp = old + fminf(fmaxf(dv, -vclip), vclip);
        pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
      }
      const float e = vp - R;
      dout[0] = vcoef * 2.f * e * inv_nb * pass;
      rs[0] = e * e;
    }
  }
  __syncthreads();
  if (j < 16) DOUT[(size_t)row * 16 + j] = dout[j];
  // dH2 = Wh^T dOut, dz2 = dH2 (1 - h2^2); dH1 = W2^T dz2, dz1 = dH1 (1 - h1^2)
  {
    const int Wh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
    float s = 0.f;
    for (int o = 0; o < n_out; ++o) s = fmaf(P[Wh + (size_t)o * L.H2 + j], dout[o], s);
    const float h2 = sh.h2[role][j];
    const float dz2 = fmaf(-(h2 * h2), s, s);
    DZ2[(size_t)row * HP + j] = dz2;
    __syncthreads();
    sh.h2[role][j] = dz2;         // (h2 is no longer needed in LDS)
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < L.H2; ++k) t = fmaf(P[L.W2[role] + (size_t)k * L.H1 + j], sh.h2[role][k], t);
    const float h1 = sh.h1[role][j];
    DZ1[(size_t)row * HP + j] = fmaf(-(h1 * h1), t, t);
  }
}

// one thread per parameter: its gradient = the sum over the minibatch rows, in row order; block partial of the squared norm
__global__ void __launch_bounds__(256) gen_wgrad_kernel(GenArgs a, int nb) {
  __shared__ float red[256];
  const GenCtl* ctl = reinterpret_cast<const GenCtl*>(a.scratch);
  if (ctl->stop) return;
  const PolLayout& L = a.L;
  const int HP = a.HP, B = a.B, O = L.O;
  const int e = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  if (e < L.n) {
    const int* rowidx = reinterpret_cast<const int*>(a.scratch + gen_off_rowidx(B));
    if (!L.discrete && e < L.A) {          // log_std: sum_rows dlp (dd^2 / var - 1), and d(ent_coef * -mean H) / d log_std = -ent_coef
      const float* g2 = a.scratch + gen_off_g2(B);
      for (int r = 0; r < nb; ++r) g += g2[(size_t)r * 16 + e];
      g += -a.hp.ent_coef;
    } else {
      int role = -1, kind = 0, off = 0;    // kind 0 W1, 1 b1, 2 W2, 3 b2, 4 head weight, 5 head bias
      for (int w = 0; w < 3; ++w) {
        if (e >= L.W1[w] && e < L.b1[w]) { role = w; kind = 0; off = e - L.W1[w]; }
        else if (e >= L.b1[w] && e < L.W2[w]) { role = w; kind = 1; off = e - L.b1[w]; }
        else if (e >= L.W2[w] && e < L.b2[w]) { role = w; kind = 2; off = e - L.W2[w]; }
        else if (e >= L.b2[w] && e < L.b2[w] + L.H2) { role = w; kind = 3; off = e - L.b2[w]; }
      }
      if (role < 0) {
        if (e >= L.Wa && e < L.ba) { role = 0; kind = 4; off = e - L.Wa; }
        else if (e >= L.ba && e < L.Wv) { role = 0; kind = 5; off = e - L.ba; }
        else if (e >= L.Wv && e < L.bv) { role = 1; kind = 4; off = e - L.Wv; }
        else if (e == L.bv) { role = 1; kind = 5; off = 0; }
        else if (e >= L.Wc && e < L.bc) { role = 2; kind = 4; off = e - L.Wc; }
        else { role = 2; kind = 5; off = 0; }
      }
      const float* act = a.scratch + gen_off_act(B, role, HP);
      const float* H1b = act, *H2b = act + (size_t)B * HP, *DZ1 = act + (size_t)2 * B * HP, *DZ2 = act + (size_t)3 * B * HP, *DOUT = act + (size_t)4 * B * HP;
      if (kind == 0) { const int j = off / O, k = off - j * O; for (int r = 0; r < nb; ++r) g = fmaf(DZ1[(size_t)r * HP + j], a.buf.observations[(size_t)rowidx[r] * O + k], g); }
      else if (kind == 1) { for (int r = 0; r < nb; ++r) g += DZ1[(size_t)r * HP + off]; }
      else if (kind == 2) { const int j = off / L.H1, k = off - j * L.H1; for (int r = 0; r < nb; ++r) g = fmaf(DZ2[(size_t)r * HP + j], H1b[(size_t)r * HP + k], g); }
      else if (kind == 3) { for (int r = 0; r < nb; ++r) g += DZ2[(size_t)r * HP + off]; }
      else if (kind == 4) { const int o = off / L.H2, k = off - o * L.H2; for (int r = 0; r < nb; ++r) g = fmaf(DOUT[(size_t)r * 16 + o], H2b[(size_t)r * HP + k], g); }
      else { for (int r = 0; r < nb; ++r) g += DOUT[(size_t)r * 16 + off]; }
    }
    a.scratch[gen_off_grad(B, HP) + e] = g;
  }
  const float ss = block_sum(g * g, red);
  if (threadIdx.x == 0) a.scratch[gen_off_part(B, HP, L.n) + blockIdx.x] = ss;
}

// clip_grad_norm_ + torch.optim.Adam (single-tensor form) on every parameter; block 0 keeps the statistics of the step
__global__ void __launch_bounds__(256) gen_adam_kernel(GenArgs a, int step, int epoch, int mb, int nb) {
  __shared__ float coef_s;
  GenCtl* ctl = reinterpret_cast<GenCtl*>(a.scratch);
  if (ctl->stop) return;
  const PolLayout& L = a.L;
  const int HP = a.HP, B = a.B, nblk = (L.n + 255) / 256;
  if (threadIdx.x == 0) {
    const float* part = a.scratch + gen_off_part(B, HP, L.n);
    float total = 0.f;
    for (int i = 0; i < nblk; ++i) total += part[i];          // fixed order: every block forms the same total
    const float c = a.hp.max_grad_norm / (sqrtf(total) + 1e-6f);
    coef_s = c > 1.f ? 1.f : c;
  }
  __syncthreads();
  const float coef = coef_s;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < L.n) {
    const double t = (double)(a.adam_t[0] + step + 1);
    const float step_size = (float)((double)a.hp.lr / (1.0 - pow((double)a.hp.adam_beta1, t)));
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)a.hp.adam_beta2, t)));
    const float g = a.scratch[gen_off_grad(B, HP) + e] * coef;
    const float b1 = a.hp.adam_beta1, b2 = a.hp.adam_beta2;
    const float m = fmaf((float)(1.0 - (double)b1), g, b1 * a.exp_avg[e]);
    const float v = fmaf((float)(1.0 - (double)b2), g * g, b2 * a.exp_avg_sq[e]);
    a.exp_avg[e] = m; a.exp_avg_sq[e] = v;
    a.params[e] = fmaf(-step_size, m / fmaf(sqrtf(v), inv_bc2_sqrt, a.hp.adam_eps), a.params[e]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float* rs = a.scratch + gen_off_rowstat();
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, vr = 0.f, vc = 0.f;
    for (int r = 0; r < nb; ++r) {
      const float* q = rs + (size_t)r * 8;
      s0 += q[0]; s1 += q[1]; s2 += q[2]; s3 += q[3]; s4 += q[4];
      vr += rs[((size_t)B + r) * 8]; vc += rs[((size_t)2 * B + r) * 8];
    }
    const float inv_nb = 1.f / (float)nb, nu = a.nu[0];
    const float ent = s4 * inv_nb;        // continuous: every row carries the same sum over log_std
    const float entropy_loss = -ent;
    const float pl = (-(s0 * inv_nb) + nu * (s1 * inv_nb)) / (1.f + nu);
    ctl->acc_ent += entropy_loss; ctl->acc_pg += pl; ctl->acc_cf += s2 * inv_nb;
    ctl->acc_vl_r += vr * inv_nb; ctl->acc_vl_c += vc * inv_nb;
    ctl->last_pol = pl + a.hp.ent_coef * entropy_loss; ctl->last_vl_r = vr * inv_nb; ctl->last_vl_c = vc * inv_nb;
    if (mb == 0) ctl->kl_acc = 0.f;
    ctl->kl_acc += s3 * inv_nb;
    ctl->steps_done += 1;
    if (mb == a.n_mb - 1) {
      const float mean_kl = ctl->kl_acc / (float)a.n_mb;
      ctl->mean_kl = mean_kl;
      a.stats[32 + epoch] = mean_kl;
      if (a.hp.use_target_kl && mean_kl > 1.5f * a.hp.target_kl) { ctl->stop = 1; ctl->early_stop_epoch = epoch; }
    }
  }
}

__global__ void gen_finish_kernel(GenArgs a, int* adam_t) {
  const GenCtl* ctl = reinterpret_cast<const GenCtl*>(a.scratch);
  a.stats[0] = ctl->stop ? (float)ctl->early_stop_epoch : (float)a.hp.n_epochs;
  a.stats[1] = (float)ctl->steps_done;
  a.stats[2] = ctl->acc_ent; a.stats[3] = ctl->acc_pg; a.stats[4] = ctl->acc_vl_r; a.stats[5] = ctl->acc_vl_c; a.stats[6] = ctl->acc_cf;
  a.stats[7] = ctl->mean_kl; a.stats[8] = ctl->last_pol; a.stats[9] = ctl->last_vl_r; a.stats[10] = ctl->last_vl_c; a.stats[11] = 0.f;
  adam_t[0] += ctl->steps_done;
}

long long generic_train_bytes(int batch_size, int hp, int n_params) { return (long long)gen_floats(batch_size, hp, n_params) * 4; }
static_assert(ICRL_PPO_GENERIC_BYTES(64, 128, 1000) == 4 * (64 + 64 * (24 + 1 + 16 + 3 * (4 * 128 + 16)) + 1000 + 4 + 64), "ICRL_PPO_GENERIC_BYTES");

// perm_off: the permutations already mapped to storage offsets (prepare in ppo_train.hip); scratch: ICRL_PPO_GENERIC_BYTES
int launch_train_generic(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, const icrl_buffer_t* buf,
                         const int* perm_off, const float* nu, const icrl_ppo_hyper_t* hp, float* stats, void* scratch, hipStream_t s) {
  if (pol->h1 != pol->h2 || pol->h1 % 64 != 0 || pol->h1 > GEN_MAX_H || pol->obs_dim > 1024 || pol->act_dim > MAX_ACT)
    return fail("icrl_ppo_lag_train (generic path): obs_dim %d (<= 1024), act_dim %d (<= %d), padded hidden width %d x %d (equal, a multiple of 64, "
                "<= %d)", pol->obs_dim, pol->act_dim, MAX_ACT, pol->h1, pol->h2, GEN_MAX_H);
  GenArgs a;
  a.L = make_pol_layout(pol->obs_dim, pol->act_dim, pol->h1, pol->h2, pol->discrete);
  a.HP = pol->h1; a.B = hp->batch_size;
  a.params = pol->params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.adam_t = adam_step;
  a.buf = *buf; a.perm_off = perm_off; a.nu = nu; a.hp = *hp; a.stats = stats; a.scratch = (float*)scratch;
  a.n_total = buf->T * buf->N;
  a.n_mb = (a.n_total + hp->batch_size - 1) / hp->batch_size;
  if ((size_t)generic_train_bytes(a.B, a.HP, a.L.n) != ICRL_PPO_GENERIC_BYTES(a.B, a.HP, a.L.n)) return fail("generic update: scratch layout and ICRL_PPO_GENERIC_BYTES disagree");
  hipError_t e = hipMemsetAsync(scratch, 0, 64 * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  const int nblk = (a.L.n + 255) / 256;
  int step = 0;
  for (int ep = 0; ep < hp->n_epochs; ++ep)
    for (int mb = 0; mb < a.n_mb; ++mb, ++step) {
      const int p0 = mb * hp->batch_size;
      const int nb = a.n_total - p0 < hp->batch_size ? a.n_total - p0 : hp->batch_size;
      const int base = ep * a.n_total + p0;
      hipLaunchKernelGGL(gen_stats_kernel, dim3(1), dim3(256), 0, s, a, base, nb);
      hipLaunchKernelGGL(gen_forward_backward_kernel, dim3(nb, 3), dim3(a.HP), 0, s, a, base, nb);
      hipLaunchKernelGGL(gen_wgrad_kernel, dim3(nblk), dim3(256), 0, s, a, nb);
      hipLaunchKernelGGL(gen_adam_kernel, dim3(nblk), dim3(256), 0, s, a, step, ep, mb, nb);
    }
  hipLaunchKernelGGL(gen_finish_kernel, dim3(1), dim3(1), 0, s, a, adam_step);
  return (int)hipGetLastError();
}

}  // namespace icrl
