// Generic-shape path of the policy kernels — gfx950.
//
// The fast kernels (rollout.hip, ppo_train_*.hip) are built around three separate two-layer branches of 64-wide hidden layers,
// minibatches of at most 256 rows and one lane per hidden unit.  The reference accepts any `-sl / -pl / -rvl / -cvl` layer lists and
// any batch size (icrl/utils.py:636-655, stable_baselines3/common/torch_layers.py:129-254, common/buffers.py:594-612); this file
// serves what the fast kernels refuse, from ONE table of layers (GenNet):
//   * two-layer branches up to 256 units in the classic layout (icrl_policy_t.arch == NULL; stored padded to a common width h1 = h2 =
//     a multiple of 64: pad units have zero weights and biases, output tanh(0) = 0 and receive zero gradients, exactly like the narrow
//     widths of the fast kernels),
//   * any MlpExtractor architecture (icrl_policy_t.arch != NULL): a shared trunk of 0..4 layers, then 0..4 layers per branch, every
//     width 1..256, natural (unpadded) parameter layout in the reference's state_dict order,
//   * minibatches of any size.
// Plain kernels, one thread per hidden unit / per parameter, sequential fmaf chains, several launches per optimiser step — correct
// and deterministic, not latency-tuned: every BASELINE config runs on the fast kernels.
//
//   icrl_policy_forward / icrl_policy_evaluate      -> policy_generic_kernel            (policies.py:716-731, 752-767)
//   icrl_ppo_lag_train                              -> per optimiser step: gen_forward_backward | gen_wgrad_tiled | gen_adam (+ the next step's gen_stats)
//                                                      (ppo_lag.py:196-299, torch.optim.Adam, clip_grad_norm_)
// The forward reads the weights from `params_t` = the per-layer transposes (icrl_policy_prepare; the update keeps both images current).
#include "ppo_common.h"
#include "generic.h"
#ifndef GEN_ROWS_AHEAD
#define GEN_ROWS_AHEAD 8      // rows whose operands the weight-gradient kernel fetches ahead of their dependent fmas
#endif

namespace icrl {

struct GenCtl {      // first 64 floats of the generic scratch, zeroed at the start of a train() launch sequence
  float mean_r, istd_r, mean_c;
  int stop, steps_done, early_stop_epoch;      // stop: 0, or 1 + the index of the optimiser step that decided the target-KL early stop
  float kl_acc;
  float acc_ent, acc_pg, acc_cf, acc_vl_r, acc_vl_c, last_pol, last_vl_r, last_vl_c, mean_kl;
};

struct GenArgs {
  int B;                     // batch_size
  float* params; float* params_t; float* exp_avg; float* exp_avg_sq;      // params_t: kept equal to the transposes of params by gen_adam_kernel
  const int* adam_t;
  icrl_buffer_t buf;
  const int* perm_off;       // [n_epochs * T*N] storage offsets (ppo_perm_offsets_kernel)
  const float* nu;
  icrl_ppo_hyper_t hp;
  float* stats;
  float* scratch;            // ICRL_PPO_GENERIC_BYTES
  int n_total, n_mb;
};

// scratch layout in floats (RF = GenNet.row_floats): control | per-row loss terms | row -> storage offset | log_std gradient terms |
// activations [B][RF] | pre-activation gradients [B][RF] | gradient [n] | per-block partial squared norms
__host__ __device__ inline size_t gen_off_rowstat() { return 64; }
__host__ __device__ inline size_t gen_off_rowidx(int B) { return gen_off_rowstat() + (size_t)3 * B * 8; }
__host__ __device__ inline size_t gen_off_g2(int B) { return gen_off_rowidx(B) + (size_t)B; }
__host__ __device__ inline size_t gen_off_act(int B) { return gen_off_g2(B) + (size_t)B * 16; }
__host__ __device__ inline size_t gen_off_dz(int B, int RF) { return gen_off_act(B) + (size_t)B * RF; }
__host__ __device__ inline size_t gen_off_grad(int B, int RF) { return gen_off_dz(B, RF) + (size_t)B * RF; }
__host__ __device__ inline size_t gen_off_part(int B, int RF, int n_params) { return gen_off_grad(B, RF) + (size_t)n_params; }
__host__ __device__ inline size_t gen_floats(int B, int RF, int n_params) { return gen_off_part(B, RF, n_params) + (size_t)(n_params + 255) / 256 + 1088; }

// position of parameter e in the transposed image: weights W[j][k] -> Wt[k][j] inside their layer's block, everything else in place
__device__ __forceinline__ int gen_transposed_index(const GenNet& net, int e) {
  for (int l = 0; l < net.n_layers; ++l) {
    const GenLayer& y = net.layer[l];
    if (e >= y.w_off && e < y.b_off) {
      const int off = e - y.w_off, j = off / y.in_dim, k = off - j * y.in_dim;
      return y.w_off + k * y.out_dim + j;
    }
  }
  return e;
}

__global__ void __launch_bounds__(256) gen_transpose_kernel(GenNet net, const float* __restrict__ P, float* __restrict__ PT) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < net.n) PT[gen_transposed_index(net, e)] = P[e];
}

// policies.py:716-731 (forward: sample / deterministic, clip, log-prob) and :752-767 (evaluate_actions: `given` actions, entropy)
__global__ void __launch_bounds__(3 * GEN_MAX_H) policy_generic_kernel(GenNet net, const float* __restrict__ P, const float* __restrict__ PT, const double* __restrict__ obs,
                                                                        const float* __restrict__ noise, int deterministic, const float* alow,
                                                                        const float* ahigh, float* actions, float* act_clipped, float* v_r,
                                                                        float* v_c, float* log_prob, const float* __restrict__ given,
                                                                        float* entropy) {
  __shared__ GenFwdShared sh;
  const int tid = threadIdx.x, slot = tid / net.W, j = tid - slot * net.W;
  const size_t n = blockIdx.x;
  for (int i = tid; i < net.O; i += blockDim.x) sh.x[i] = (float)obs[n * net.O + i];
  __syncthreads();
  gen_mlp_forward(net, PT, sh.x, sh.act, slot, j);
  if (tid == 0) {
    const int AS = net.discrete ? 1 : net.A;
    float lp, ent;
    gen_policy_head(net, P, sh.act + net.layer[net.head[0]].act_off, noise ? noise + n * AS : nullptr, deterministic, alow, ahigh,
                    given ? given + n * AS : nullptr, actions ? actions + n * AS : nullptr, act_clipped ? act_clipped + n * AS : nullptr, lp, ent);
    if (v_r) v_r[n] = sh.act[net.layer[net.head[1]].act_off];
    if (v_c) v_c[n] = sh.act[net.layer[net.head[2]].act_off];
    if (log_prob) log_prob[n] = lp;
    if (entropy) entropy[n] = ent;
  }
}

int launch_policy_generic(const icrl_policy_t* p, const double* obs, const float* noise, int N, int deterministic, const float* alow,
                          const float* ahigh, float* actions, float* act_clipped, float* v_r, float* v_c, float* log_prob,
                          const float* given, float* entropy, hipStream_t s) {
  GenNet net;
  if (int e = make_gen_net(p, &net, "policy forward / evaluate")) return e;
  if (p->params_t == nullptr) return fail("policy forward / evaluate (generic path): params_t is NULL (icrl_policy_prepare fills it)");
  hipLaunchKernelGGL(policy_generic_kernel, dim3(N), dim3(3 * net.W), 0, s, net, p->params, p->params_t, obs, noise, deterministic, alow, ahigh, actions,
                     act_clipped, v_r, v_c, log_prob, given, entropy);
  return (int)hipGetLastError();
}

int policy_generic_check(const icrl_policy_t* p, const char* who) {
  GenNet net;
  return make_gen_net(p, &net, who);
}

// icrl_policy_prepare of a generic-path policy: params_t = the per-layer transposes the forward reads
int launch_generic_transpose(const icrl_policy_t* p, hipStream_t s) {
  GenNet net;
  if (int e = make_gen_net(p, &net, "icrl_policy_prepare")) return e;
  if (p->params_t == nullptr) return fail("icrl_policy_prepare: params_t is NULL");
  hipLaunchKernelGGL(gen_transpose_kernel, dim3((net.n + 255) / 256), dim3(256), 0, s, net, p->params, p->params_t);
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
__device__ __forceinline__ void gen_stats_body(const GenArgs& a, int perm_base, int nb, float* red) {
  GenCtl* ctl = reinterpret_cast<GenCtl*>(a.scratch);
  float sr = 0.f, sc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { const int idx = a.perm_off[perm_base + i]; sr += a.buf.reward_advantages[idx]; sc += a.buf.cost_advantages[idx]; }
  const float mean_r = block_sum(sr, red) / (float)nb, mean_c = block_sum(sc, red) / (float)nb;
  float ss = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { const float d = a.buf.reward_advantages[a.perm_off[perm_base + i]] - mean_r; ss += d * d; }
  const float var = block_sum(ss, red) / (float)(nb - 1);
  if (threadIdx.x == 0) { ctl->mean_r = mean_r; ctl->mean_c = mean_c; ctl->istd_r = 1.f / (sqrtf(var) + 1e-8f); }
}

// (the first step's; every later step's statistics are formed by the last workgroup of the previous step's gen_adam_kernel)
__global__ void __launch_bounds__(256) gen_stats_kernel(GenArgs a, int perm_base, int nb) {
  __shared__ float red[256];
  if (reinterpret_cast<const GenCtl*>(a.scratch)->stop) return;
  gen_stats_body(a, perm_base, nb, red);
}

// forward, loss and activation backward of ONE minibatch row through the whole network: grid nb, block 3 * W
__global__ void __launch_bounds__(3 * GEN_MAX_H) gen_forward_backward_kernel(GenNet net, GenArgs a, int perm_base, int nb) {
  __shared__ GenFwdShared sh;
  __shared__ float dz[GEN_MAX_ROW];      // d loss / d pre-activation of every layer of the row, laid out like sh.act
  const GenCtl* ctl = reinterpret_cast<const GenCtl*>(a.scratch);
  if (ctl->stop) return;
  const int B = a.B, RF = net.row_floats, tid = threadIdx.x, slot = tid / net.W, j = tid - slot * net.W, row = blockIdx.x;
  const float* P = a.params;
  const int idx = a.perm_off[perm_base + row];
  for (int i = tid; i < net.O; i += blockDim.x) sh.x[i] = a.buf.observations[(size_t)idx * net.O + i];
  __syncthreads();
  gen_mlp_forward(net, a.params_t, sh.x, sh.act, slot, j);
  if (j == 0) {      // one thread per slot: the loss terms of its head (pi | vf | cvf)
    const int role = slot;
    float* dout = dz + net.layer[net.head[role]].act_off;
    const float nu = a.nu[0], inv_nb = 1.f / (float)nb;
    float* rs = a.scratch + gen_off_rowstat() + ((size_t)role * B + row) * 8;
    if (role == 0) {
      reinterpret_cast<int*>(a.scratch + gen_off_rowidx(B))[row] = idx;
      float* g2row = a.scratch + gen_off_g2(B) + (size_t)row * 16;
      const float* out = sh.act + net.layer[net.head[0]].act_off;
      const int A = net.A;
      float lp = 0.f, ent = 0.f, g1[MAX_ACT], g2[MAX_ACT];
      if (!net.discrete) {
        for (int o = 0; o < A; ++o) {
          const float ls = P[net.log_std + o], sd = __expf(ls), iv = 1.f / (sd * sd);
          const float dd = a.buf.actions[(size_t)idx * a.buf.act_store + o] - out[o];
          lp += -(dd * dd) * (0.5f * iv) - ls - LOG_SQRT_2PI_F;
          g1[o] = dd * iv;
          g2[o] = (dd * dd) * iv - 1.f;
          ent += HALF_LOG_2PI_PLUS_HALF_F + ls;
        }
      } else {
        float m = -INFINITY;
        for (int o = 0; o < A; ++o) m = fmaxf(m, out[o]);
        float se = 0.f;
        for (int o = 0; o < A; ++o) se += expf(out[o] - m);
        const float lse = m + logf(se);
        const int action = (int)a.buf.actions[(size_t)idx * a.buf.act_store];
        for (int o = 0; o < A; ++o) { const float lg = out[o] - lse; ent -= expf(lg) * lg; }
        for (int o = 0; o < A; ++o) {
          const float lg = out[o] - lse, pr = expf(lg);
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
      for (int o = 0; o < 16; ++o) g2row[o] = 0.f;
      for (int o = 0; o < A; ++o) {
        dout[o] = net.discrete ? dlp * g1[o] + a.hp.ent_coef * inv_nb * g2[o] : dlp * g1[o];
        g2row[o] = net.discrete ? 0.f : dlp * g2[o];
      }
      rs[0] = fminf(s1, s2); rs[1] = Ac * ratio; rs[2] = fabsf(ratio - 1.f) > clip ? 1.f : 0.f; rs[3] = old_lp - lp; rs[4] = ent;
    } else if (role < 3) {
      const float v = sh.act[net.layer[net.head[role]].act_off];
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
  // back through the stages: d h = sum over the layers that read h of W^T dz (in layer order), dz = d h (1 - h^2)
  for (int s = net.n_stages - 2; s >= 0; --s) {
    const int l = gen_my_layer(net, s, slot);
    if (l >= 0 && j < net.layer[l].out_dim) {
      float t = 0.f;
      for (int c = l + 1; c < net.n_layers; ++c) {
        const GenLayer& y = net.layer[c];
        if (y.in_buf != l) continue;
        const float* w = P + y.w_off + j;
        const float* d = dz + y.act_off;
        const int n_o = y.out_dim, n_i = y.in_dim;
        int i = 0;
        for (; i + GEN_AHEAD <= n_o; i += GEN_AHEAD) {      // (GEN_AHEAD weights in flight before their dependent fmas; same chain order)
          float wv[GEN_AHEAD];
#pragma unroll
          for (int u = 0; u < GEN_AHEAD; ++u) wv[u] = w[(size_t)(i + u) * n_i];
#pragma unroll
          for (int u = 0; u < GEN_AHEAD; ++u) t = fmaf(wv[u], d[i + u], t);
        }
        for (; i < n_o; ++i) t = fmaf(w[(size_t)i * n_i], d[i], t);
      }
      const float h = sh.act[net.layer[l].act_off + j];
      dz[net.layer[l].act_off + j] = fmaf(-(h * h), t, t);
    }
    __syncthreads();
  }
  float* ACT = a.scratch + gen_off_act(B) + (size_t)row * RF;
  float* DZ = a.scratch + gen_off_dz(B, RF) + (size_t)row * RF;
  for (int i = tid; i < RF; i += blockDim.x) { ACT[i] = sh.act[i]; DZ[i] = dz[i]; }
}

// one thread per parameter: its gradient = the sum over the minibatch rows, in row order; block partial of the squared norm
__global__ void __launch_bounds__(256) gen_wgrad_kernel(GenNet net, GenArgs a, int nb) {
  __shared__ float red[256];
  const GenCtl* ctl = reinterpret_cast<const GenCtl*>(a.scratch);
  if (ctl->stop) return;
  const int B = a.B, RF = net.row_floats, O = net.O;
  const int e = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  if (e < net.n) {
    const int* rowidx = reinterpret_cast<const int*>(a.scratch + gen_off_rowidx(B));
    if (!net.discrete && e < net.A) {          // log_std: sum_rows dlp (dd^2 / var - 1), and d(ent_coef * -mean H) / d log_std = -ent_coef
      const float* g2 = a.scratch + gen_off_g2(B);
      for (int r = 0; r < nb; ++r) g += g2[(size_t)r * 16 + e];
      g += -a.hp.ent_coef;
    } else {
      int l = 0;
      while (l < net.n_layers - 1 && !(e >= net.layer[l].w_off && e < net.layer[l].b_off + net.layer[l].out_dim)) ++l;
      const GenLayer& y = net.layer[l];
      const float* ACT = a.scratch + gen_off_act(B);
      const float* DZ = a.scratch + gen_off_dz(B, RF) + y.act_off;
      if (e < y.b_off) {                       // W[j][k]: sum_rows dz[j] * input[k]
        const int off = e - y.w_off, j = off / y.in_dim, k = off - j * y.in_dim;
        // (eight rows' operands are fetched before the eight dependent fmas: the chain's order is the row order either way)
        int r = 0;
        if (y.in_buf < 0) {
          const float* ob = a.buf.observations + k;
          for (; r + GEN_ROWS_AHEAD <= nb; r += GEN_ROWS_AHEAD) {
            float d[GEN_ROWS_AHEAD], x[GEN_ROWS_AHEAD];
#pragma unroll
            for (int u = 0; u < GEN_ROWS_AHEAD; ++u) { d[u] = DZ[(size_t)(r + u) * RF + j]; x[u] = ob[(size_t)rowidx[r + u] * O]; }
#pragma unroll
            for (int u = 0; u < GEN_ROWS_AHEAD; ++u) g = fmaf(d[u], x[u], g);
          }
          for (; r < nb; ++r) g = fmaf(DZ[(size_t)r * RF + j], ob[(size_t)rowidx[r] * O], g);
        } else {
          const float* in = ACT + net.layer[y.in_buf].act_off + k;
          for (; r + GEN_ROWS_AHEAD <= nb; r += GEN_ROWS_AHEAD) {
            float d[GEN_ROWS_AHEAD], x[GEN_ROWS_AHEAD];
#pragma unroll
            for (int u = 0; u < GEN_ROWS_AHEAD; ++u) { d[u] = DZ[(size_t)(r + u) * RF + j]; x[u] = in[(size_t)(r + u) * RF]; }
#pragma unroll
            for (int u = 0; u < GEN_ROWS_AHEAD; ++u) g = fmaf(d[u], x[u], g);
          }
          for (; r < nb; ++r) g = fmaf(DZ[(size_t)r * RF + j], in[(size_t)r * RF], g);
        }
      } else {                                 // b[j]
        const int j = e - y.b_off;
        int r = 0;
        for (; r + GEN_ROWS_AHEAD <= nb; r += GEN_ROWS_AHEAD) {
          float d[GEN_ROWS_AHEAD];
#pragma unroll
          for (int u = 0; u < GEN_ROWS_AHEAD; ++u) d[u] = DZ[(size_t)(r + u) * RF + j];
#pragma unroll
          for (int u = 0; u < GEN_ROWS_AHEAD; ++u) g += d[u];
        }
        for (; r < nb; ++r) g += DZ[(size_t)r * RF + j];
      }
    }
    a.scratch[gen_off_grad(B, RF) + e] = g;
  }
  const float ss = block_sum(g * g, red);
  if (threadIdx.x == 0) a.scratch[gen_off_part(B, RF, net.n) + blockIdx.x] = ss;
}

// The same gradients by 16 x 16 parameter tiles: workgroup = one tile W[16 jt .., 16 kt ..] of one layer (thread (tj, tk) = one
// parameter; the threads tk = 0 of the tiles kt = 0 also carry the bias b[16 jt + tj]), the rows' dz and inputs staged through LDS
// 64 rows at a time — every staged value is read 16 times from LDS instead of once per parameter from L2.  Each parameter still adds
// its rows in row order (same chain as gen_wgrad_kernel: bit-identical gradients); workgroup 0 does log_std.  Partial squared norms:
// one per workgroup (a different grouping than gen_wgrad_kernel's 256 consecutive parameters: the clip coefficient can differ in the
// last bit).
constexpr int GEN_TILE = 16, GEN_TROWS = 64;
__host__ __device__ inline int gen_tiles_of(const GenLayer& y) { return ((y.out_dim + GEN_TILE - 1) / GEN_TILE) * ((y.in_dim + GEN_TILE - 1) / GEN_TILE); }

__global__ void __launch_bounds__(256) gen_wgrad_tiled_kernel(GenNet net, GenArgs a, int nb) {
  __shared__ float red[256];
  __shared__ float dzs[GEN_TROWS][GEN_TILE + 1], ins[GEN_TROWS][GEN_TILE + 1];
  const GenCtl* ctl = reinterpret_cast<const GenCtl*>(a.scratch);
  if (ctl->stop) return;
  const int B = a.B, RF = net.row_floats, O = net.O, tid = threadIdx.x;
  float* grad = a.scratch + gen_off_grad(B, RF);
  float g = 0.f, gb = 0.f;
  if (blockIdx.x == 0) {      // log_std: sum_rows dlp (dd^2 / var - 1), and d(ent_coef * -mean H) / d log_std = -ent_coef
    if (!net.discrete && tid < net.A) {
      const float* g2 = a.scratch + gen_off_g2(B) + tid;
      int r = 0;
      for (; r + 16 <= nb; r += 16) {      // (sixteen rows' terms in flight before their dependent adds: one load per iteration made this
        float v[16];                       // workgroup the kernel's longest at large batches)
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = g2[(size_t)(r + u) * 16];
#pragma unroll
        for (int u = 0; u < 16; ++u) g += v[u];
      }
      for (; r < nb; ++r) g += g2[(size_t)r * 16];
      g += -a.hp.ent_coef;
      grad[tid] = g;
    }
  } else {
    int t = (int)blockIdx.x - 1, l = 0;
    while (l < net.n_layers - 1 && t >= gen_tiles_of(net.layer[l])) { t -= gen_tiles_of(net.layer[l]); ++l; }
    const GenLayer& y = net.layer[l];
    const int ktiles = (y.in_dim + GEN_TILE - 1) / GEN_TILE, jt = t / ktiles, kt = t - jt * ktiles;
    const int tj = tid >> 4, tk = tid & 15, j = jt * GEN_TILE + tj, k = kt * GEN_TILE + tk;
    const int* rowidx = reinterpret_cast<const int*>(a.scratch + gen_off_rowidx(B));
    const float* ACT = a.scratch + gen_off_act(B);
    const float* DZ = a.scratch + gen_off_dz(B, RF) + y.act_off;
    const float* IN = y.in_buf < 0 ? nullptr : ACT + net.layer[y.in_buf].act_off;
    const bool bias = kt == 0 && tk == 0;
    for (int r0 = 0; r0 < nb; r0 += GEN_TROWS) {
      // stage 64 rows x 16 columns of dz (columns 16 jt ..) and of the layer's input (columns 16 kt ..): four values of each per thread, all in flight together
      for (int i = tid; i < GEN_TROWS * GEN_TILE; i += 256) {
        const int rr = i >> 4, c = i & 15, r = r0 + rr;
        float dv = 0.f, xv = 0.f;
        if (r < nb) {
          if (jt * GEN_TILE + c < y.out_dim) dv = DZ[(size_t)r * RF + jt * GEN_TILE + c];
          if (kt * GEN_TILE + c < y.in_dim)
            xv = IN == nullptr ? a.buf.observations[(size_t)rowidx[r] * O + kt * GEN_TILE + c] : IN[(size_t)r * RF + kt * GEN_TILE + c];
        }
        dzs[rr][c] = dv; ins[rr][c] = xv;
      }
      __syncthreads();
      const int nr = nb - r0 < GEN_TROWS ? nb - r0 : GEN_TROWS;
      for (int rr = 0; rr < nr; ++rr) {
        const float d = dzs[rr][tj];
        g = fmaf(d, ins[rr][tk], g);
        gb += d;
      }
      __syncthreads();
    }
    if (j < y.out_dim && k < y.in_dim) grad[y.w_off + (size_t)j * y.in_dim + k] = g; else g = 0.f;
    if (bias && j < y.out_dim) grad[y.b_off + j] = gb; else gb = 0.f;
  }
  const float ss = block_sum(fmaf(g, g, gb * gb), red);
  if (tid == 0) a.scratch[gen_off_part(B, RF, net.n) + blockIdx.x] = ss;
}

// clip_grad_norm_ + torch.optim.Adam (single-tensor form) on every parameter (both images: params and its per-layer transposes);
// block 0 keeps the statistics of the step.  Sums are fixed trees over the block (block_sum): every block forms the same total.
__global__ void __launch_bounds__(256) gen_adam_kernel(GenNet net, GenArgs a, int n_parts, int step, int epoch, int mb, int nb, int next_base, int next_nb) {
  __shared__ float red[256];
  GenCtl* ctl = reinterpret_cast<GenCtl*>(a.scratch);
  // The stop word holds 1 + the step that decided it, and only LATER steps leave: workgroup 0 of this very launch may write it below
  // while other workgroups (or waves of one workgroup) are still being dispatched, and the reference applies the epoch's last
  // optimizer.step() in full before it breaks (ppo_lag.py:286-297) — a plain flag would let late workgroups skip their share of it.
  const int stop_at = ctl->stop;
  if (stop_at != 0 && stop_at <= step) return;
  const int B = a.B, RF = net.row_floats, n_params = net.n, nblk = n_parts, tid = threadIdx.x;      // n_parts: partial squared norms the gradient kernel left
  float coef;
  {
    const float* part = a.scratch + gen_off_part(B, RF, n_params);
    float p = 0.f;
    for (int i = tid; i < nblk; i += 256) p += part[i];
    const float total = block_sum(p, red);
    const float c = a.hp.max_grad_norm / (sqrtf(total) + 1e-6f);
    coef = c > 1.f ? 1.f : c;
  }
  // bias corrections in double, once per block (two pow() calls per thread were a third of this kernel's time)
  __shared__ float bc_s[2];
  if (tid == 0) {
    const double t = (double)(a.adam_t[0] + step + 1);
    bc_s[0] = (float)((double)a.hp.lr / (1.0 - pow((double)a.hp.adam_beta1, t)));
    bc_s[1] = (float)(1.0 / sqrt(1.0 - pow((double)a.hp.adam_beta2, t)));
  }
  __syncthreads();
  const int e = blockIdx.x * 256 + tid;
  if (e < n_params) {
    const float step_size = bc_s[0], inv_bc2_sqrt = bc_s[1];
    const float g = a.scratch[gen_off_grad(B, RF) + e] * coef;
    const float b1 = a.hp.adam_beta1, b2 = a.hp.adam_beta2;
    const float m = fmaf((float)(1.0 - (double)b1), g, b1 * a.exp_avg[e]);
    const float v = fmaf((float)(1.0 - (double)b2), g * g, b2 * a.exp_avg_sq[e]);
    a.exp_avg[e] = m; a.exp_avg_sq[e] = v;
    const float w = fmaf(-step_size, m / fmaf(sqrtf(v), inv_bc2_sqrt, a.hp.adam_eps), a.params[e]);
    a.params[e] = w;
    a.params_t[gen_transposed_index(net, e)] = w;
  }
  if (blockIdx.x == 0) {
    const float* rs = a.scratch + gen_off_rowstat();
    float q[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < nb; r += 256) {
      const float* x = rs + (size_t)r * 8;
      q[0] += x[0]; q[1] += x[1]; q[2] += x[2]; q[3] += x[3]; q[4] += x[4];
      q[5] += rs[((size_t)B + r) * 8]; q[6] += rs[((size_t)2 * B + r) * 8];
    }
    for (int i = 0; i < 7; ++i) q[i] = block_sum(q[i], red);
    if (tid == 0) {
      const float s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3], s4 = q[4], vr = q[5], vc = q[6];
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
        if (a.hp.use_target_kl && mean_kl > 1.5f * a.hp.target_kl) { ctl->stop = step + 1; ctl->early_stop_epoch = epoch; }
      }
    }
  }
  // the advantage statistics of the NEXT minibatch (read by the next step's forward / backward launch): the last workgroup, behind its
  // share of the update — one launch less per step.  (The statistics in `ctl` were last read by this step's forward / backward.)  This
  // workgroup runs whenever the next step will: a launch only leaves early for a stop decided by an EARLIER launch (stop_at above).
  if (next_nb > 0 && blockIdx.x == gridDim.x - 1) {
    __syncthreads();
    gen_stats_body(a, next_base, next_nb, red);
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

// =====================================================================================================================
// ONE persistent launch per train() (round 5): the same GenNet table, the update as fp32 MFMA tiles.
// =====================================================================================================================
// grid = G = ceil(batch_size / 16) workgroups of 8 waves; workgroup g takes rows 16 g .. 16 g + 15 of EVERY minibatch through the whole
// network — activations and pre-activation gradients of its 16 rows live in LDS, row-major and padded per layer to 16 units; every GEMM
// is a chain of v_mfma_f32_16x16x4_f32 whose A operand (weights: `params` row-major for the forward and the weight gradients' layout,
// the per-layer transposes `params_t` for the backward) is streamed from L2 and whose B operand is one ds_read_b128 of the row image;
// the 16 x 16 tiles of a stage are dealt round-robin to the waves — then writes its PARTIAL weight gradients (K = its 16 rows: four
// MFMAs per parameter tile) to `part[g]`.  Across workgroups, per optimiser step: grid barrier | workgroup g sums the partials of ITS
// parameter slice over the tiles in tile order (deterministic), leaves the gradient and its squared norm | grid barrier | every
// workgroup forms the same total norm and runs clip + Adam on its slice (both parameter images) | grid barrier.  Data that crosses
// workgroups (partials, parameters, norms, loss sums) is written and read with agent-scope relaxed atomics (sc1: no stale L1 / non-local L2
// line), a barrier arrival is preceded by s_waitcnt vmcnt(0); no fences.  Every workgroup accumulates the SAME logged sums and takes
// the same target-KL decision from them, so nothing has to be broadcast.
// Shapes this form does not serve (more than GENP_MAX_TILES row tiles, more than GENP_MAX_PARAMS parameters, rows that do not fit the
// LDS) keep the three-launches-per-step form below.
constexpr int GENP_TH = 512;
constexpr int GENP_MAX_TILES = 32;            // batch_size <= 512
constexpr int GENP_MIN_TILES = 16;            // the default from this many row tiles on (batch_size > 240): below, the launch-per-phase form is faster (measured)
constexpr int GENP_MAX_PARAMS = 131072;
constexpr int GENP_MAX_RS = 1096;             // floats per LDS row: 2 images x 16 rows x RS x 4 B <= 140 KB

struct GenPersist {
  int RS, OB, G, H, slice;                    // LDS row stride | observation width padded to 16 | row-tile workgroups | all workgroups (G + helpers of the reduce / Adam phases) | parameters per workgroup there
  int poff[GEN_MAX_LAYERS];                   // padded offset of each layer's output inside a row (the observation sits at 0)
  const PlanStep* plan;                       // per optimiser step: Adam's bias corrections, rows, flags, permutation base
  int n_steps;
  float* part; float* grad; float* norm; float* stat; unsigned* bar; int* tidx;      // [G][n] | [n] | [G] | [G][8] | barrier counter | [n] position of a parameter in params_t
};

// loads: agent scope (sc1: the L1 is bypassed; served by the XCD's L2 when the line is there).  stores: agent scope (sc1: written through, the
// line is DROPPED from the L2 — every later load pays the fabric, ~2 us) unless all workgroups of the launch share one XCD (`local`,
// checked at run time: HW_REG_XCC_ID of every workgroup), then workgroup scope (sc0: the line stays in the one L2 all of them read through)
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool LOCAL>
__device__ __forceinline__ void st_x(float* p, float v) {
  if (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Shared read-mostly data (parameters, partial gradients) goes through buffer resources with the sc1 cache policy — L1 bypassed,
// served by the XCD's L2 — as plain (non-atomic) loads: a chain of relaxed ATOMIC loads is issued one at a time (each waited for
// before the next: 16 trips to the L2 per K chunk, measured 6 k cycles per chunk), buffer loads are scheduled freely, and an offset
// beyond the resource returns 0, which is the mask of the padded tiles.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t genp_rsrc(const float* p, size_t n_floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n_floats * 4), 0x00020000);
}
// cache policy of the loads: sc1 (agent scope) is only served by the L2 for a line that is DIRTY there — anything else goes to the fabric
// (~6 k cycles measured) —; when all workgroups share an XCD (`local`) sc0 is enough: the L1 is bypassed and the one L2 everybody
// writes through serves every resident line (~700 cycles)
template <bool LOCAL>
__device__ __forceinline__ float genp_ld(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {      // (the policy is an immediate of the instruction: a compile-time choice)
  if (LOCAL) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 1));
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 16));
}
constexpr unsigned GENP_OOB = 0x80000000u;      // an offset outside every resource (they hold <= 16 MB), with room for the constant offsets added to it: the load returns 0, the store is dropped

// acc[i] (unit 16 t + 4 q + i, row r) += sum_k A[16 t + r][k] B[r][k]:  A[m][k] = W[w_off + m * ldk + k] (zero outside nM x nK),
// B = the LDS row image at `brow` (= image + r * RS + the input's padded offset).  K in chunks of 64 (16 weights per lane), the next
// chunk's loads in flight under this chunk's MFMAs.
template <bool LOCAL>
__device__ __forceinline__ f32x4 genp_gemm(f32x4 acc, __amdgpu_buffer_rsrc_t rs, int w_off, int t, int ldk, int nM, int nK, const float* brow, int r, int q) {
  // Masking without a branch in the load stream (a select between two LOADS becomes divergent control flow with a wait in every arm —
  // measured 10 k cycles per tile): a row m >= nM starts at an offset outside the resource, so all its loads return 0; columns
  // k >= nK need no mask at all — the B operand's pad columns are zeros (every tile is stored 16 wide with its rows masked like this),
  // the weights read there are some other finite parameters, or 0 past the end of the resource.
  const int m = 16 * t + r;
  const unsigned base = m < nM ? 4u * (unsigned)(w_off + m * ldk + 4 * q) : GENP_OOB;
  auto fetch = [&](int k0, float (&a)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * c + e] = genp_ld<LOCAL>(rs, base + 4u * (unsigned)(k0 + 16 * c + e));
  };
  float cur[16], nxt[16];
  fetch(0, cur);
  for (int k0 = 0; k0 < nK; k0 += 64) {
    const bool more = k0 + 64 < nK;
    if (more) fetch(k0 + 64, nxt);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (k0 + 16 * c < nK) {
        const f32x4 b = lds128(brow + k0 + 16 * c + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = MFMA_F32(cur[4 * c + e], b[e], acc);
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
    }
  }
  return acc;
}

template <bool LOCAL>
__device__ __forceinline__ void genp_grid_barrier(unsigned* bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    if (LOCAL) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (atomics execute in the L2: the one all workgroups share)
    else __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// fixed tree over the 8 waves of the workgroup (wave sums by DPP, then the eight in order): every workgroup forms the same value
__device__ __forceinline__ float genp_block_sum(float v, float* red8) {
  v = wave_sum_fast(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red8[w] = v;
  __syncthreads();
  return ((red8[0] + red8[1]) + (red8[2] + red8[3])) + ((red8[4] + red8[5]) + (red8[6] + red8[7]));
}

// the body for one store / load policy (LOCAL: every workgroup of the launch sits on the same XCD)
template <bool LOCAL>
__device__ __forceinline__ void genp_body(const GenNet& net, const GenArgs& a, const GenPersist& pp, const int g, float* const sm, unsigned bar_n) {
  constexpr bool local = LOCAL;
  const int RS = pp.RS, G = pp.G, H = pp.H;
  const bool tile_wg = g < G;                  // workgroups G .. H-1 only help with the reduce / Adam phases
  float* const ACT = sm;                       // [16][RS] observation + every layer's output
  float* const DZ = sm + 16 * RS;              // [16][RS] d loss / d pre-activation
  float* const G2 = DZ + 16 * RS;              // [16][16] log_std gradient terms of the rows
  float* const RST = G2 + 256;                 // [3][16][8] per-row loss terms
  float* const RED = RST + 384;                // [16] reduction scratch
  int* const RIDX = reinterpret_cast<int*>(RED + 16);      // [16] storage offsets of the rows
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int n = net.n, O = net.O, A = net.A;
  const __amdgpu_buffer_rsrc_t rsP = genp_rsrc(a.params, (size_t)n), rsPT = genp_rsrc(a.params_t, (size_t)n), rsPart = genp_rsrc(pp.part, (size_t)G * n);
  const int lo = g * pp.slice, hi = (lo + pp.slice < n) ? lo + pp.slice : n;      // this workgroup's parameter slice
  // replicated logged state (identical in every workgroup)
  float acc_ent = 0.f, acc_pg = 0.f, acc_cf = 0.f, acc_vr = 0.f, acc_vc = 0.f, last_pol = 0.f, last_vr = 0.f, last_vc = 0.f, kl_acc = 0.f, mean_kl = 0.f;
  int steps_done = 0, early_stop_epoch = a.hp.n_epochs;
  const float nu = a.nu[0];
  bool stop = false;
  const bool prof = (a.hp._pad & 1) != 0 && g == 0;       // phase timers of workgroup 0, thread 0 (tools/generic_only.py PROF=1): stats[12..21]
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = prof ? stamp() : 0ull;
#define GSTAMP(k) if (prof) { const unsigned long long now_ = stamp(); ph[k] += now_ - t_last; t_last = now_; }
  for (int st = 0; st < pp.n_steps && !stop; ++st) {
    const PlanStep ps = pp.plan[st];
    const int nb = ps.nb_flags & NB_MASK, epoch = ps.nb_flags >> NB_EPOCH;
    const bool first_mb = (ps.nb_flags >> NB_FIRST) & 1, last_mb = (ps.nb_flags >> NB_LAST) & 1;
    const float inv_nb = 1.f / (float)nb;
    if (tile_wg) {
    // ---- rows of this tile -> LDS; advantage statistics of the whole minibatch (every workgroup, identically)
    if (tid < 16) {
      const int row = 16 * g + tid;
      RIDX[tid] = a.perm_off[ps.perm_base + (row < nb ? row : 0)];
    }
    float sr = 0.f, sc = 0.f, srr = 0.f;
    for (int i = tid; i < nb; i += GENP_TH) {
      const int idx = a.perm_off[ps.perm_base + i];
      const float ar = a.buf.reward_advantages[idx];
      sr += ar; sc += a.buf.cost_advantages[idx]; srr += ar * ar;
    }
    __syncthreads();
    for (int i = tid; i < 16 * pp.OB; i += GENP_TH) {
      const int rr = i / pp.OB, k = i - rr * pp.OB;
      ACT[rr * RS + k] = k < O ? a.buf.observations[(size_t)RIDX[rr] * O + k] : 0.f;
    }
    sr = genp_block_sum(sr, RED); sc = genp_block_sum(sc, RED); srr = genp_block_sum(srr, RED);
    const float mean_r = sr * inv_nb, mean_c = sc * inv_nb;
    const float istd_r = 1.f / (sqrtf(fmaxf(srr - sr * mean_r, 0.f) / (float)(nb - 1)) + 1e-8f);
    __syncthreads();
    GSTAMP(0)   // rows + advantage statistics
    // ================= forward, stage by stage =================
    for (int s = 0; s < net.n_stages; ++s) {
      int item = 0;
      for (int l = net.stage_begin[s]; l < net.stage_begin[s + 1]; ++l) {
        const GenLayer& y = net.layer[l];
        const int in_off = y.in_buf < 0 ? 0 : pp.poff[y.in_buf];
        for (int t = 0; t < (y.out_dim + 15) / 16; ++t, ++item) {
          if ((item & 7) != w) continue;
          f32x4 acc;
#pragma unroll
          for (int i = 0; i < 4; ++i) { const int j = 16 * t + 4 * q + i; acc[i] = genp_ld<LOCAL>(rsP, j < y.out_dim ? 4u * (unsigned)(y.b_off + j) : GENP_OOB); }
          acc = genp_gemm<LOCAL>(acc, rsP, y.w_off, t, y.in_dim, y.out_dim, y.in_dim, ACT + r * RS + in_off, r, q);
          if (y.tanh) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fast_tanh(acc[i]);
          }
          *reinterpret_cast<f32x4*>(ACT + r * RS + pp.poff[l] + 16 * t + 4 * q) = acc;
        }
      }
      __syncthreads();
    }
    GSTAMP(1)   // forward
    // ================= loss terms of the three heads: thread (role, row) =================
    if (tid < 48) {
      const int role = tid >> 4, row = tid & 15;
      const bool valid = 16 * g + row < nb;
      const int idx = RIDX[row];
      float* dout = DZ + row * RS + pp.poff[net.head[role]];
      const float* out = ACT + row * RS + pp.poff[net.head[role]];
      float* rs = RST + (role * 16 + row) * 8;
      if (role == 0) {
        float lp = 0.f, ent = 0.f, g1[MAX_ACT], g2[MAX_ACT];
        if (!net.discrete) {
          for (int o = 0; o < A; ++o) {
            const float ls = genp_ld<LOCAL>(rsP, 4u * (unsigned)(net.log_std + o)), sd = __expf(ls), iv = 1.f / (sd * sd);
            const float dd = a.buf.actions[(size_t)idx * a.buf.act_store + o] - out[o];
            lp += -(dd * dd) * (0.5f * iv) - ls - LOG_SQRT_2PI_F;
            g1[o] = dd * iv;
            g2[o] = (dd * dd) * iv - 1.f;
            ent += HALF_LOG_2PI_PLUS_HALF_F + ls;
          }
        } else {
          float m = -INFINITY;
          for (int o = 0; o < A; ++o) m = fmaxf(m, out[o]);
          float se = 0.f;
          for (int o = 0; o < A; ++o) se += expf(out[o] - m);
          const float lse = m + logf(se);
          const int action = (int)a.buf.actions[(size_t)idx * a.buf.act_store];
          for (int o = 0; o < A; ++o) { const float lg = out[o] - lse; ent -= expf(lg) * lg; }
          for (int o = 0; o < A; ++o) {
            const float lg = out[o] - lse, pr = expf(lg);
            if (o == action) lp = lg;
            g1[o] = (o == action ? 1.f : 0.f) - pr;
            g2[o] = pr * (lg + ent);
          }
        }
        const float old_lp = a.buf.log_probs[idx];
        const float ratio = __expf(lp - old_lp);
        const float Ar = (a.buf.reward_advantages[idx] - mean_r) * istd_r;
        const float Ac = a.buf.cost_advantages[idx] - mean_c;
        const float clip = a.hp.clip_range;
        const float s1 = Ar * ratio, s2 = Ar * fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const float gsel = (s1 <= s2) ? Ar : 0.f;
        const float dlp = valid ? inv_nb / (1.f + nu) * (-gsel + nu * Ac) * ratio : 0.f;
        const float dent = valid ? a.hp.ent_coef * inv_nb : 0.f;
        for (int o = 0; o < 16; ++o) G2[row * 16 + o] = 0.f;
        for (int o = 0; o < A; ++o) {
          dout[o] = net.discrete ? dlp * g1[o] + dent * g2[o] : dlp * g1[o];
          G2[row * 16 + o] = net.discrete ? 0.f : dlp * g2[o];
        }
        rs[0] = valid ? fminf(s1, s2) : 0.f; rs[1] = valid ? Ac * ratio : 0.f; rs[2] = (valid && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
        rs[3] = valid ? old_lp - lp : 0.f; rs[4] = valid ? ent : 0.f;
      } else {
        const float v = out[0];
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
        dout[0] = valid ? vcoef * 2.f * e * inv_nb * pass : 0.f;
        rs[0] = valid ? e * e : 0.f;
      }
    }
    __syncthreads();
    GSTAMP(2)   // loss
    // ================= backward of the activations, stage by stage: d h = sum over the layers that read h of W^T dz (layer order) =================
    for (int s = net.n_stages - 2; s >= 0; --s) {
      int item = 0;
      for (int l = net.stage_begin[s]; l < net.stage_begin[s + 1]; ++l) {
        const GenLayer& y = net.layer[l];
        for (int t = 0; t < (y.out_dim + 15) / 16; ++t, ++item) {
          if ((item & 7) != w) continue;
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int c = l + 1; c < net.n_layers; ++c) {
            const GenLayer& z = net.layer[c];
            if (z.in_buf != l) continue;
            acc = genp_gemm<LOCAL>(acc, rsPT, z.w_off, t, z.out_dim, z.in_dim, z.out_dim, DZ + r * RS + pp.poff[c], r, q);      // A[k][j] = Wt[k * out + j]
          }
          const f32x4 h = lds128(ACT + r * RS + pp.poff[l] + 16 * t + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = fmaf(-(h[i] * h[i]), acc[i], acc[i]);
          *reinterpret_cast<f32x4*>(DZ + r * RS + pp.poff[l] + 16 * t + 4 * q) = acc;
        }
      }
      __syncthreads();
    }
    GSTAMP(3)   // backward
    // ================= partial weight gradients of this tile's 16 rows -> part[g] =================
    float* const mypart = pp.part + (size_t)g * n;
    {
      int item = 0;
      for (int l = 0; l < net.n_layers; ++l) {
        const GenLayer& y = net.layer[l];
        const int in_off = y.in_buf < 0 ? 0 : pp.poff[y.in_buf];
        for (int jt = 0; jt < (y.out_dim + 15) / 16; ++jt, ++item) {
          if ((item & 7) != w) continue;
          float az[4];      // dz[row 4 q + e][unit 16 jt + r]
#pragma unroll
          for (int e = 0; e < 4; ++e) az[e] = DZ[(4 * q + e) * RS + pp.poff[l] + 16 * jt + r];
          unsigned jrow[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const int j = 16 * jt + 4 * q + i; jrow[i] = j < y.out_dim ? 4u * (unsigned)(g * n + y.w_off + j * y.in_dim) : GENP_OOB / 2; }
          for (int kt = 0; kt < (y.in_dim + 15) / 16; ++kt) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = MFMA_F32(az[e], ACT[(4 * q + e) * RS + in_off + 16 * kt + r], acc);
            // (an element outside the layer: row or column offset outside the resource — two halves of GENP_OOB, so that one or both
            // of them push the sum out of range without wrapping — and the store is dropped; no branch in the store stream)
            const unsigned koff = 16 * kt + r < y.in_dim ? 4u * (unsigned)(16 * kt + r) : GENP_OOB / 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned off = jrow[i] + koff;
              if (LOCAL) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i]), rsPart, (int)off, 0, 1);
              else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i]), rsPart, (int)off, 0, 16);
            }
          }
          // bias: the sum over the 16 rows (lanes r = unit, q = row group): rows 4 q .. 4 q + 3, then over q
          const float sb = quad_rows_sum((az[0] + az[1]) + (az[2] + az[3]));
          if (q == 0 && 16 * jt + r < y.out_dim) st_x<LOCAL>(mypart + y.b_off + 16 * jt + r, sb);
        }
      }
      if (!net.discrete && tid < A) {
        float sl = 0.f;
        for (int rr = 0; rr < 16; ++rr) sl += G2[rr * 16 + tid];
        st_x<LOCAL>(mypart + net.log_std + tid, sl);
      }
      if (tid < 8) {      // this tile's loss sums: policy terms 0..4, reward / cost value errors 5, 6
        float v = 0.f;
        if (tid < 5) for (int rr = 0; rr < 16; ++rr) v += RST[rr * 8 + tid];
        else if (tid < 7) for (int rr = 0; rr < 16; ++rr) v += RST[((tid - 4) * 16 + rr) * 8];
        st_x<LOCAL>(pp.stat + g * 8 + tid, v);
      }
    }
    }      // tile_wg
    GSTAMP(4)   // weight gradients
    bar_n += H; genp_grid_barrier<LOCAL>(pp.bar, bar_n);           // (A) every tile's partials are in memory
    GSTAMP(5)   // barrier A
    // ================= this workgroup's parameter slice: sum over the tiles in tile order, squared norm =================
    float ss = 0.f;
    for (int e0 = lo + 4 * tid; e0 < hi; e0 += 4 * GENP_TH) {      // four consecutive elements per thread: G 16-byte loads in flight
      f32x4 gs = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < G; t += 4) {
        f32x4 pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned off = t + u < G ? 4u * (unsigned)((t + u) * n + e0) : GENP_OOB;
          pv[u] = __builtin_bit_cast(f32x4, LOCAL ? __builtin_amdgcn_raw_buffer_load_b128(rsPart, (int)off, 0, 1) : __builtin_amdgcn_raw_buffer_load_b128(rsPart, (int)off, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) gs[i] += pv[u][i];      // tile order (absent tiles read 0)
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = e0 + i;
        if (e < hi) {
          if (!net.discrete && e >= net.log_std && e < net.log_std + A) gs[i] += -a.hp.ent_coef;      // d(ent_coef * -mean H) / d log_std
          pp.grad[e] = gs[i];
          ss = fmaf(gs[i], gs[i], ss);
        }
      }
    }
    ss = genp_block_sum(ss, RED);
    if (tid == 0) st_x<LOCAL>(pp.norm + g, ss);
    GSTAMP(6)   // reduce + norm
    bar_n += H; genp_grid_barrier<LOCAL>(pp.bar, bar_n);           // (B) every slice's squared norm is in memory
    GSTAMP(7)   // barrier B
    float total = 0.f, q7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {   // every tile's loss sums and every slice's squared norm: fetched side by side (one value per thread; a chain of relaxed atomic
        // loads, or a loop of dependent buffer loads, is one L2 round trip per value: 26 k cycles at 32 tiles), summed in tile order from LDS
      const __amdgpu_buffer_rsrc_t rsX = genp_rsrc(pp.stat, 512);      // [stat G x 8 | norm H]
      __syncthreads();
      if (tid < 8 * G) RST[tid] = genp_ld<LOCAL>(rsX, 4u * (unsigned)tid);
      else if (tid >= 256 && tid < 256 + H) RST[tid] = genp_ld<LOCAL>(rsX, 4u * (unsigned)tid);
      __syncthreads();
      for (int t = 0; t < H; ++t) total += RST[256 + t];
      for (int t = 0; t < G; ++t) {
#pragma unroll
        for (int k = 0; k < 7; ++k) q7[k] += RST[t * 8 + k];
      }
    }
    const float cc = a.hp.max_grad_norm / (sqrtf(total) + 1e-6f), coef = cc > 1.f ? 1.f : cc;
    {
      const float b1 = a.hp.adam_beta1, b2 = a.hp.adam_beta2, w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
      for (int e0 = lo + 4 * tid; e0 < hi; e0 += 4 * GENP_TH) {
        float gv[4], mv[4], vv[4], pw[4];
        int ti[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ec = e0 + u < hi ? e0 + u : lo;
          gv[u] = pp.grad[ec]; mv[u] = a.exp_avg[ec]; vv[u] = a.exp_avg_sq[ec]; pw[u] = genp_ld<LOCAL>(rsP, 4u * (unsigned)ec); ti[u] = pp.tidx[ec];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u;
          if (e < hi) {
            const float gc = gv[u] * coef;
            const float m = fmaf(w1, gc, b1 * mv[u]);
            const float v = fmaf(w2, gc * gc, b2 * vv[u]);
            a.exp_avg[e] = m; a.exp_avg_sq[e] = v;
            const float nw = fmaf(-ps.step_size, m / fmaf(sqrtf(v), ps.inv_bc2_sqrt, a.hp.adam_eps), pw[u]);
            st_x<LOCAL>(a.params + e, nw);
            st_x<LOCAL>(a.params_t + ti[u], nw);
          }
        }
      }
    }
    {   // logged sums (every workgroup, identically; workgroup 0 writes them out at the end)
      const float ent = q7[4] * inv_nb, entropy_loss = -ent;
      const float pl = (-(q7[0] * inv_nb) + nu * (q7[1] * inv_nb)) / (1.f + nu);
      acc_ent += entropy_loss; acc_pg += pl; acc_cf += q7[2] * inv_nb;
      acc_vr += q7[5] * inv_nb; acc_vc += q7[6] * inv_nb;
      last_pol = pl + a.hp.ent_coef * entropy_loss; last_vr = q7[5] * inv_nb; last_vc = q7[6] * inv_nb;
      if (first_mb) kl_acc = 0.f;
      kl_acc += q7[3] * inv_nb;
      ++steps_done;
      if (last_mb) {
        mean_kl = kl_acc / (float)a.n_mb;
        if (g == 0 && tid == 0) a.stats[32 + epoch] = mean_kl;
        if (a.hp.use_target_kl && mean_kl > 1.5f * a.hp.target_kl) { stop = true; early_stop_epoch = epoch; }
      }
    }
    GSTAMP(8)   // Adam + logged sums
    bar_n += H; genp_grid_barrier<LOCAL>(pp.bar, bar_n);           // (C) the updated parameters are in memory
    GSTAMP(9)   // barrier C
  }
  if (prof && tid == 0) a.stats[22] = local ? 1.f : 0.f;
  if (prof && tid == 0)
    for (int k = 0; k < 10; ++k) a.stats[12 + k] = (float)((double)ph[k] / (double)(steps_done > 0 ? steps_done : 1));
  if (g == 0 && tid == 0) {
    a.stats[0] = (float)early_stop_epoch;
    a.stats[1] = (float)steps_done;
    a.stats[2] = acc_ent; a.stats[3] = acc_pg; a.stats[4] = acc_vr; a.stats[5] = acc_vc; a.stats[6] = acc_cf;
    a.stats[7] = mean_kl; a.stats[8] = last_pol; a.stats[9] = last_vr; a.stats[10] = last_vc; a.stats[11] = 0.f;
    const_cast<int*>(a.adam_t)[0] += steps_done;
  }
}

__global__ void __launch_bounds__(GENP_TH) gen_train_persistent_kernel(GenNet net, GenArgs a, GenPersist pp, int packed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // packed: a 1-D grid of 8 (G - 1) + 1 workgroups of which every eighth works — workgroups are dealt round-robin over the 8 XCDs, so
  // the G working ones land on ONE (ppo_common.h: XCD placement); the others leave at once
  if (packed && (blockIdx.x & (XCD_STRIDE - 1)) != 0) return;
  const int H = pp.H, g = packed ? (int)blockIdx.x / XCD_STRIDE : (int)blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 32 * pp.RS + 256 + 384 + 32; i += GENP_TH) sm[i] = 0.f;
  for (int e = g * pp.slice + tid; e < g * pp.slice + pp.slice && e < net.n; e += GENP_TH) pp.tidx[e] = gen_transposed_index(net, e);      // (read back by the same thread)
  __syncthreads();
  // do all workgroups share an XCD?  (every workgroup publishes its XCC id in its norm slot, one barrier, everybody compares)
  unsigned bar_n = 0;
  if (tid == 0) __hip_atomic_store(pp.norm + g, __uint_as_float(0x100u | xcc_id()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bar_n += H; genp_grid_barrier<false>(pp.bar, bar_n);
  bool local = true;
  for (int t = 0; t < H; ++t) local = local && __float_as_uint(ld_sc1(pp.norm + t)) == (0x100u | xcc_id());
  local = __builtin_amdgcn_readfirstlane((int)local) != 0;
  bar_n += H; genp_grid_barrier<false>(pp.bar, bar_n);      // (the slots are reused by the first step's norms)
  if (local) genp_body<true>(net, a, pp, g, sm, bar_n);
  else genp_body<false>(net, a, pp, g, sm, bar_n);
}

static_assert(ICRL_PPO_GENERIC_BYTES(64, 776, 1000) == 4 * (64 + 64 * (24 + 1 + 16 + 2 * 776) + 1000 + 4 + 1088 + 5 * 1000 + 1024), "ICRL_PPO_GENERIC_BYTES");
static_assert(ICRL_PPO_GENERIC_BYTES(1024, 776, 1000) == 4 * (64 + 1024 * (24 + 1 + 16 + 2 * 776) + 1000 + 4 + 1088), "ICRL_PPO_GENERIC_BYTES");

// the schedule table of ppo_train.hip (per optimiser step: Adam's bias corrections in double, rows, flags, permutation base)
__global__ void ppo_plan_kernel(const int* adam_t, int n_steps, int n_mb, int n_total, int B, double lr, double b1, double b2, PlanStep* steps,
                                PlanChunk* chunks, int two_per_step);

// the persistent form, when the shape allows it: 0 launched | < 0 not eligible (the caller uses the launch-per-phase form) | > 0 error
static int launch_train_generic_persistent(const GenNet& net, GenArgs& a, const icrl_ppo_hyper_t* hp, int32_t* adam_step, void* sync_ws, hipStream_t s) {
  // WHERE IT IS THE DEFAULT IS MEASURED (round 5, MI355X, tools/generic_only.py): a 64-row minibatch is FOUR row tiles, i.e. four compute
  // units walking every stage of the network one after the other (three 16 x 16 tiles per wave and stage, each a 32-MFMA chain behind an
  // L2 round trip), where the launch-per-phase form spreads the same rows over 64 compute units: -pl 128 128 -rvl 128 128 -cvl 128 128 at
  // batch 64: 56 us per optimiser step against 45 (per step of workgroup 0, cycles: rows + statistics 4.9 k | forward 29.2 k | loss 6.0 k |
  // backward 24.6 k | weight gradients 36.4 k | barrier 2.0 k | reduce 2.6 k | barrier 2.7 k | Adam + logged sums 13.5 k | barrier 8.4 k);
  // with 32 row tiles (default widths, batch 512) the same phases take 15.2 | 11.6 | 17.3 k and the step 48 us against 64.  So: the default
  // from GENP_MIN_TILES row tiles on; ICRL_GEN_PERSISTENT=1 / ICRL_GEN_LAUNCHES=1 force one form (tests run both).
  static const bool on = getenv("ICRL_GEN_PERSISTENT") != nullptr, off = getenv("ICRL_GEN_LAUNCHES") != nullptr;
  const int B = hp->batch_size, G = (B + 15) / 16, n = net.n;
  if (off || (!on && G < GENP_MIN_TILES) || G > GENP_MAX_TILES || n > GENP_MAX_PARAMS || B > NB_MASK) return -1;
  GenPersist pp;
  pp.OB = (net.O + 15) / 16 * 16;
  int off_ = pp.OB;
  for (int l = 0; l < net.n_layers; ++l) { pp.poff[l] = off_; off_ += (net.layer[l].out_dim + 15) / 16 * 16; }
  pp.RS = off_ + ((8 - off_ % 32) + 32) % 32;      // = 8 mod 32: conflict-free ds_read_b128 of a row image
  if (pp.RS > GENP_MAX_RS) return -1;
  const long long n_steps = (long long)hp->n_epochs * a.n_mb;
  if (n_steps >= (1ll << 21)) return -1;
  // helpers for the reduce / Adam phases: ~2048 parameters per workgroup, all workgroups on one XCD when that fits (<= its CUs)
  int H = (n + 2047) / 2048;
  H = H < G ? G : (H > 30 ? (G > 30 ? G : 30) : H);
  pp.G = G; pp.H = H; pp.slice = ((n + H - 1) / H + 3) / 4 * 4;
  PlanStep* steps = reinterpret_cast<PlanStep*>((char*)sync_ws + 512);
  hipLaunchKernelGGL(ppo_plan_kernel, dim3((unsigned)((n_steps + 2 + 255) / 256)), dim3(256), 0, s, adam_step, (int)n_steps, a.n_mb, a.n_total, B,
                     (double)hp->lr, (double)hp->adam_beta1, (double)hp->adam_beta2, steps, (PlanChunk*)nullptr, 0);
  pp.plan = steps; pp.n_steps = (int)n_steps;
  float* extra = a.scratch + gen_floats(B, net.row_floats, n);      // behind the launch-per-phase layout: [stat G x 8 | norm G | part G x n]
  pp.stat = extra; pp.norm = extra + 256; pp.tidx = reinterpret_cast<int*>(extra + 512); pp.part = extra + 512 + n;
  pp.grad = a.scratch + gen_off_grad(B, net.row_floats);
  pp.bar = reinterpret_cast<unsigned*>(a.scratch + 60);           // (the first 64 floats are zeroed by the caller)
  const size_t lds = (size_t)(32 * pp.RS + 256 + 384 + 32) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)gen_train_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gen_transpose_kernel, dim3((n + 255) / 256), dim3(256), 0, s, net, a.params, a.params_t);
  GenNet net_ = net; GenArgs a_ = a;
  static const bool nopack = getenv("ICRL_NO_XCD_PACK") != nullptr;
  int packed = nopack ? 0 : 1;
  void* params[] = {(void*)&net_, (void*)&a_, (void*)&pp, (void*)&packed};
  e = hipLaunchCooperativeKernel((const void*)gen_train_persistent_kernel, dim3(packed ? XCD_STRIDE * (H - 1) + 1 : H), dim3(GENP_TH), params, (unsigned)lds, s);
  if (e == hipErrorCooperativeLaunchTooLarge && packed) {      // (a partition that cannot hold the sparse grid: the dense one, agent-scope stores)
    (void)hipGetLastError();
    packed = 0;
    e = hipLaunchCooperativeKernel((const void*)gen_train_persistent_kernel, dim3(H), dim3(GENP_TH), params, (unsigned)lds, s);
  }
  return (int)e;
}

// perm_off: the permutations already mapped to storage offsets (prepare in ppo_train.hip); scratch: ICRL_PPO_GENERIC_BYTES
int launch_train_generic(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, const icrl_buffer_t* buf,
                         const int* perm_off, const float* nu, const icrl_ppo_hyper_t* hp, float* stats, void* scratch, void* sync_ws, hipStream_t s) {
  GenNet net;
  if (int e = make_gen_net(pol, &net, "icrl_ppo_lag_train")) return e;
  GenArgs a;
  a.B = hp->batch_size;
  if (pol->params_t == nullptr) return fail("icrl_ppo_lag_train (generic path): params_t is NULL");
  a.params = pol->params; a.params_t = pol->params_t; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.adam_t = adam_step;
  a.buf = *buf; a.perm_off = perm_off; a.nu = nu; a.hp = *hp; a.stats = stats; a.scratch = (float*)scratch;
  a.n_total = buf->T * buf->N;
  a.n_mb = (a.n_total + hp->batch_size - 1) / hp->batch_size;
  if ((gen_floats(a.B, net.row_floats, net.n) + ICRL_PPO_GENERIC_PERSIST_FLOATS(a.B, net.n)) * 4 != ICRL_PPO_GENERIC_BYTES(a.B, net.row_floats, net.n)) return fail("generic update: scratch layout and ICRL_PPO_GENERIC_BYTES disagree");
  hipError_t e = hipMemsetAsync(scratch, 0, 64 * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  {
    const int pe = launch_train_generic_persistent(net, a, hp, adam_step, sync_ws, s);
    if (pe >= 0) return pe;
  }
  const int nblk = (net.n + 255) / 256;
  int n_tiles = 1;      // workgroups of gen_wgrad_tiled_kernel: log_std + the 16 x 16 tiles of every layer
  for (int l = 0; l < net.n_layers; ++l) n_tiles += gen_tiles_of(net.layer[l]);
  static const bool flat = getenv("ICRL_GEN_WGRAD_FLAT") != nullptr;      // A/B: one thread per parameter straight from L2 (gen_wgrad_kernel)
  const bool tiled = !flat && n_tiles <= nblk + 1024;                      // (partial-norm slots of the scratch)
  hipLaunchKernelGGL(gen_transpose_kernel, dim3(nblk), dim3(256), 0, s, net, pol->params, pol->params_t);
  int step = 0;
  for (int ep = 0; ep < hp->n_epochs; ++ep)
    for (int mb = 0; mb < a.n_mb; ++mb, ++step) {
      const int p0 = mb * hp->batch_size;
      const int nb = a.n_total - p0 < hp->batch_size ? a.n_total - p0 : hp->batch_size;
      const int base = ep * a.n_total + p0;
      if (step == 0) hipLaunchKernelGGL(gen_stats_kernel, dim3(1), dim3(256), 0, s, a, base, nb);
      hipLaunchKernelGGL(gen_forward_backward_kernel, dim3(nb), dim3(3 * net.W), 0, s, net, a, base, nb);
      if (tiled) hipLaunchKernelGGL(gen_wgrad_tiled_kernel, dim3(n_tiles), dim3(256), 0, s, net, a, nb);
      else hipLaunchKernelGGL(gen_wgrad_kernel, dim3(nblk), dim3(256), 0, s, net, a, nb);
      // (the next step's minibatch: position p0 + batch_size of this epoch, or the start of the next epoch)
      const bool last = ep == hp->n_epochs - 1 && mb == a.n_mb - 1;
      const int np0 = mb + 1 < a.n_mb ? p0 + hp->batch_size : 0, nep = mb + 1 < a.n_mb ? ep : ep + 1;
      const int next_nb = last ? 0 : (a.n_total - np0 < hp->batch_size ? a.n_total - np0 : hp->batch_size);
      hipLaunchKernelGGL(gen_adam_kernel, dim3(nblk), dim3(256), 0, s, net, a, tiled ? n_tiles : nblk, step, ep, mb, nb, last ? 0 : nep * a.n_total + np0, next_nb);
    }
  hipLaunchKernelGGL(gen_finish_kernel, dim3(1), dim3(1), 0, s, a, adam_step);
  return (int)hipGetLastError();
}

// floats one row's activation record takes (the `row_floats` argument of ICRL_PPO_GENERIC_BYTES), or -1 (reason in icrl_last_error)
int generic_row_floats(const icrl_policy_t* pol) {
  GenNet net;
  if (make_gen_net(pol, &net, "icrl_ppo_generic_row_floats")) return -1;
  return net.row_floats;
}

}  // namespace icrl
