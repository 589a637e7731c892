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
// Workgroups of 8 waves, all on ONE XCD (<= 30 of its 32 compute units).  A TILE workgroup (g, b) takes rows 16 g .. 16 g + 15 of every
// minibatch through the layers of branch b (NBR = 3: policy / value / cost-value workgroups per row tile, every one of them also runs
// the shared trunk forward; chosen while 3 G <= 30) or through the whole table (NBR = 1).  Activations and pre-activation gradients of
// its 16 rows live in LDS, row-major, every layer padded to 16 units; every GEMM is a chain of v_mfma_f32_16x16x4_f32:
//   forward    Z^T tile (16 units x 16 rows): A = 16 weight rows streamed from the XCD's L2 (one tile's whole K per wave, the NEXT tile
//              of the wave's work list — also across stages — in flight under this tile's MFMAs), B = ds_read_b128 of the row image;
//   backward   d h (16 rows x 64 units per item): A = ds_read_b128 of the consumers' d z, B = the consumers' weights read in their
//              NATURAL layout — one 16-byte load serves four output tiles whose columns are interleaved (unit 64 T + 4 r + i belongs to
//              tile i), so no transposed parameter image is needed; the consumers' units are split over the waves (K split), partial
//              sums meet in an LDS scratch;
//   gradients  d W^T tiles (16 inputs x 16 units, K = the 16 rows), four input tiles per item, 16-byte stores into part[g].
// A trunk under NBR = 3: the three workgroups of a row tile leave their branch's share of d h(trunk) in memory, the trunk's owner (the
// branch with the fewest parameters) sums them in branch order and walks the trunk backward.
// Across workgroups, per optimiser step: grid barrier | every workgroup (tile workgroups and helpers, ~2 048 parameters each) sums its
// parameter slice over the row tiles in tile order and leaves its squared norm | grid barrier | clip + Adam on the slice | grid barrier.
// The rows of the NEXT minibatch (observations, the per-row record of the loss, the advantage statistics) are fetched under the
// barriers.  Data that crosses workgroups is written with sc0 (workgroup scope: stays in the one L2 all of them read through) and read
// with sc0 buffer loads (L1 bypassed) when the run-time XCC check finds every workgroup on one XCD, else with agent scope (sc1);
// a barrier arrival is preceded by s_waitcnt vmcnt(0); no fences.  Every workgroup accumulates the SAME logged sums and takes the same
// target-KL decision from them, so nothing has to be broadcast.  `params_t` is rebuilt once, after the launch.
// Shapes this form does not serve (more than GENP_MAX_TILES row tiles, more than GENP_MAX_PARAMS parameters, rows that do not fit the
// LDS) keep the three-launches-per-step form below.
constexpr int GENP_TH = 512;
constexpr int GENP_MAX_TILES = 32;            // batch_size <= 512
constexpr int GENP_MAX_PARAMS = 131072;
constexpr int GENP_MAX_RS = 904;              // floats per LDS row: 2 images x 16 rows x RS x 4 B + what follows them <= 160 KB
constexpr int GENP_SCR = 8192;                // floats of the K-split scratch: parts x 16 rows x 64-unit groups, parts x groups <= 8
constexpr int GENP_FLIST = 64;                // forward items per wave
constexpr int GENP_BLIST = 128;               // backward items per wave
constexpr int GENP_REC = 24;                  // a row's record of the loss: actions [16] | old log-prob | adv r | adv c | return r | return c | value r | value c
constexpr int GENP_MAX_WGS = 30;
// ints of the tables: rows [16] | next rows [16] | FLN [8] | BLN [8] | SP [16] | SWD [16] | STL [16][4] | WL [32] | misc [16] | LT [32][8] | FL | BL
constexpr int GENP_I_FLN = 32, GENP_I_BLN = 40, GENP_I_SP = 48, GENP_I_SWD = 64, GENP_I_STL = 80, GENP_I_WL = 144, GENP_I_MISC = 176, GENP_I_LT = 192,
              GENP_I_FL = GENP_I_LT + 256, GENP_I_BL = GENP_I_FL + 8 * GENP_FLIST, GENP_INTS = GENP_I_BL + 8 * GENP_BLIST;
constexpr int GENP_FLOATS = GENP_SCR + 256 + 512 + 32 + 16 * GENP_REC + 16 + 16 + GENP_INTS + 32;      // LDS beyond the two row images (the last 32: phase timers)
static_assert((32 * GENP_MAX_RS + GENP_FLOATS) * 4 <= 160 * 1024, "LDS of the persistent generic-shape update");

struct GenPersist {
  int RS, OB, G, H, slice, NBR, owner, WTP;   // LDS row stride | observation width padded to 16 | row tiles | all workgroups | parameters per workgroup in the reduce / Adam phases | branches per row tile | trunk owner | padded width of the trunk's last layer
  int poff[GEN_MAX_LAYERS];                   // padded offset of each layer's output inside a row (the observation sits at 0)
  const PlanStep* plan;                       // per optimiser step: Adam's bias corrections, rows, flags, permutation base (+ 2 zero entries)
  int n_steps;
  float* part; float* grad; float* norm; float* stat; unsigned* bar; float* xt; unsigned* xflag;      // [G][n] | [n] | [H] | [G][8] | barrier counter | [G][3][16][WTP] | [G][3]
};

// The tables of one tile workgroup (LDS on the device, filled once by one thread — the step loop never indexes the kernel arguments;
// the host builds them per branch to check that they fit).
//   LT[l] = {in_dim, out_dim, w_off, b_off, in_col, out_col, flags, scol}: in_col / out_col = columns of the layer's input / output in a row image,
//           flags = tanh | backward role << 1 (0 none, 1 whole, 2 a share of the trunk's gradient, 3 share + the sum), scol = first scratch column
//   FL      forward items of a wave:  l | 16-unit tile << 5 | 128-wide K slab << 9 | last slab << 12 | stage << 13
//   BL      backward items of a wave: consumer layer | first unit / 16 << 5 | two chunks << 9 | l << 10 | 64-unit group << 15 | part << 17 | stage << 20 |
//           last of its (l, group, part) << 24 | no chunk at all << 25        (a chunk = 16 units of ONE consumer of l)
//   SP / SWD per stage: K-split parts (0: no scratch, straight into the image) / scratch row width;  STL[s] = the layers of stage s with a role (-1 ends)
//   WL      the layers whose weight gradients are formed here (-1 ends);  MISC = {head columns 0..2, stages, trunk's last layer}
struct GenpTables { int* LT; int* FL; int* FLN; int* BL; int* BLN; int* SP; int* SWD; int* STL; int* WL; int* MISC; };
__host__ __device__ inline GenpTables genp_tables(int* iw) {
  GenpTables T;
  T.FLN = iw + GENP_I_FLN; T.BLN = iw + GENP_I_BLN; T.SP = iw + GENP_I_SP; T.SWD = iw + GENP_I_SWD; T.STL = iw + GENP_I_STL; T.WL = iw + GENP_I_WL;
  T.MISC = iw + GENP_I_MISC; T.LT = iw + GENP_I_LT; T.FL = iw + GENP_I_FL; T.BL = iw + GENP_I_BL;
  return T;
}
__host__ __device__ inline bool genp_build(const GenNet& net, const int* poff, int NBR, int b, int owner, const GenpTables& T) {
  const int nsh = net.n_shared, NL = net.n_layers, NS = net.n_stages;
  bool ok = true;
  int wown[GEN_MAX_LAYERS], nwl = 0;
  for (int l = 0; l < NL; ++l) {
    const GenLayer& y = net.layer[l];
    wown[l] = (NBR == 1 || (l < nsh ? b == owner : y.slot == b)) ? 1 : 0;
    int* row = T.LT + 8 * l;
    row[0] = y.in_dim; row[1] = y.out_dim; row[2] = y.w_off; row[3] = y.b_off; row[4] = y.in_buf < 0 ? 0 : poff[y.in_buf]; row[5] = poff[l];
    row[6] = y.tanh ? 1 : 0; row[7] = 0;
    if (wown[l]) T.WL[nwl++] = l;
  }
  T.WL[nwl] = -1;
  for (int r = 0; r < 3; ++r) T.MISC[r] = poff[net.head[r]];
  T.MISC[3] = NS; T.MISC[4] = nsh - 1;
  for (int w = 0; w < 8; ++w) { T.FLN[w] = 0; T.BLN[w] = 0; }
  int rr = 0;
  for (int s = 0; s < NS; ++s)
    for (int l = net.stage_begin[s]; l < net.stage_begin[s + 1]; ++l) {
      if (!(NBR == 1 || l < nsh || net.layer[l].slot == b)) continue;
      const int nt = (net.layer[l].out_dim + 15) / 16, nkb = (net.layer[l].in_dim + 127) / 128;
      for (int t = 0; t < nt; ++t, ++rr)
        for (int kb = 0; kb < nkb; ++kb) {
          const int w = rr & 7;
          if (T.FLN[w] >= GENP_FLIST) { ok = false; continue; }
          T.FL[w * GENP_FLIST + T.FLN[w]++] = l | t << 5 | kb << 9 | (kb == nkb - 1 ? 1 : 0) << 12 | s << 13;
        }
    }
  rr = 0;
  for (int s = 0; s < 16; ++s) { T.SP[s] = 1; T.SWD[s] = 0; for (int i = 0; i < 4; ++i) T.STL[4 * s + i] = -1; }
  for (int s = NS - 2; s >= 0; --s) {
    int nsup = 0, nst = 0;
    for (int l = net.stage_begin[s]; l < net.stage_begin[s + 1]; ++l) {
      int role;
      if (NBR == 1) role = 1;
      else if (l >= nsh) role = net.layer[l].slot == b ? 1 : 0;
      else if (l == nsh - 1) role = b == owner ? 3 : 2;
      else role = b == owner ? 1 : 0;
      T.LT[8 * l + 6] |= role << 1;
      if (role) { T.LT[8 * l + 7] = 64 * nsup; nsup += (net.layer[l].out_dim + 63) / 64; T.STL[4 * s + nst++] = l; }
    }
    int P = 1;
    if (nsup > 8) P = 0; else if (nsup > 0) { while (2 * P * nsup <= 8) P *= 2; }
    T.SP[s] = P; T.SWD[s] = 64 * nsup;
    for (int i = 0; i < nst; ++i) {
      const int l = T.STL[4 * s + i];
      // the chain of chunks over the consumers of l whose d z is here, in layer order
      int total = 0;
      for (int c = l + 1; c < NL; ++c) if (net.layer[c].in_buf == l && wown[c]) total += (net.layer[c].out_dim + 15) / 16;
      const int PP = P > 0 ? P : 1, cpp = (total + PP - 1) / PP;
      for (int TT = 0; TT < (net.layer[l].out_dim + 63) / 64; ++TT)
        for (int p = 0; p < PP; ++p, ++rr) {
          const int w = rr & 7, c0 = p * cpp < total ? p * cpp : total, c1 = c0 + cpp < total ? c0 + cpp : total;
          const int common = l << 10 | TT << 15 | p << 17 | s << 20;
          if (c0 == c1) {
            if (T.BLN[w] >= GENP_BLIST) { ok = false; continue; }
            T.BL[w * GENP_BLIST + T.BLN[w]++] = common | 1 << 24 | 1 << 25;
            continue;
          }
          int ci = 0;
          for (int c = l + 1; c < NL; ++c) {
            if (!(net.layer[c].in_buf == l && wown[c])) continue;
            const int nch = (net.layer[c].out_dim + 15) / 16;
            for (int jc = 0; jc < nch; ++jc, ++ci) {
              if (ci < c0 || ci >= c1) continue;
              const bool two = jc + 1 < nch && ci + 1 < c1;      // (two chunks of the same consumer in one item)
              const bool last = ci + (two ? 2 : 1) >= c1;
              if (T.BLN[w] >= GENP_BLIST) ok = false;
              else T.BL[w * GENP_BLIST + T.BLN[w]++] = common | c | jc << 5 | (two ? 1 : 0) << 9 | (last ? 1 : 0) << 24;
              if (two) { ++jc; ++ci; }
            }
          }
        }
    }
  }
  return ok;
}

// loads: agent scope (sc1: the L1 is bypassed; served by the XCD's L2 when the line is there).  stores: agent scope (sc1: written through, the
// line is DROPPED from the L2 — every later load pays the fabric, ~2 us) unless all workgroups of the launch share one XCD (`local`,
// checked at run time: HW_REG_XCC_ID of every workgroup), then workgroup scope (sc0: the line stays in the one L2 all of them read through)
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool LOCAL>
__device__ __forceinline__ void st_x(float* p, float v) {
  if (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Shared read-mostly data (parameters, partial gradients) goes through buffer resources as plain (non-atomic) loads with a cache policy:
// a chain of relaxed ATOMIC loads is issued one at a time (each waited for before the next: 16 trips to the L2 per K chunk, measured
// 6 k cycles per chunk), buffer loads are scheduled freely; an offset beyond the resource returns 0 (checked per dword, also inside a
// 16-byte access) and a store there is dropped — the mask of the padded tiles, without a branch in the load / store stream (a select
// between two LOADS becomes divergent control flow with a wait in every arm: measured 10 k cycles per tile).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t genp_rsrc(const float* p, size_t n_floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n_floats * 4), 0x00020000);
}
// cache policy of the loads: sc1 (agent scope) is only served by the L2 for a line that is DIRTY there — anything else goes to the fabric
// (~6 k cycles measured) —; when all workgroups share an XCD (`local`) sc0 is enough: the L1 is bypassed and the one L2 everybody
// writes through serves every resident line (~700 cycles)
#ifndef GENP_LD_AUX
#define GENP_LD_AUX 16      // cache policy of the loads when every workgroup shares an XCD: 16 = sc1, 1 = sc0
#endif
template <bool LOCAL>
__device__ __forceinline__ float genp_ld(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {      // (the policy is an immediate of the instruction: a compile-time choice)
  if (LOCAL) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, GENP_LD_AUX));
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 16));
}
template <bool LOCAL>
__device__ __forceinline__ f32x4 genp_ld4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {     // (dword alignment is enough)
  if (LOCAL) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, GENP_LD_AUX));
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool LOCAL>
__device__ __forceinline__ void genp_st4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v) {
  if (LOCAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)byte_off, 0, 1);
  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)byte_off, 0, 16);
}
template <bool LOCAL>
__device__ __forceinline__ void genp_st1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float v) {
  if (LOCAL) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 1);
  else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 16);
}
constexpr unsigned GENP_OOB = 0x80000000u;      // an offset outside every resource (they hold <= 16 MB), with room for the constant offsets added to it: the load returns 0, the store is dropped

// grid barrier in two halves (work that needs nothing from the other workgroups goes between them)
template <bool LOCAL>
__device__ __forceinline__ void genp_arrive(unsigned* bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    if (LOCAL) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (atomics execute in the L2: the one all workgroups share)
    else __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// (the L1 is invalidated behind every wait: an sc0 load is a workgroup-scope load — it MAY be served by the compute unit's L1, and was, with
// the previous step's parameters, once a network was small enough to stay there: measured, test hc-bare)
// A wait gives up after GENP_SPIN_LIMIT polls (~10 s: a workgroup that never arrives — only a defect could cause that, the launch is cooperative —
// must cost a reported error, not a hung GPU) or as soon as another workgroup has given up (the word behind the counter); everybody then leaves the step
// loop and stats[11] tells the host, which raises like for the other persistent kernels.
constexpr int GENP_SPIN_LIMIT = 1 << 24;
__device__ __forceinline__ bool genp_wait(unsigned* bar, unsigned target, int* ok_lds, int limit = GENP_SPIN_LIMIT) {
  if (threadIdx.x == 0) {
    int ok = 1, spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > limit || ((spins & 1023) == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *ok_lds = ok;
  }
  __syncthreads();
  return *ok_lds != 0;
}
template <bool LOCAL>
__device__ __forceinline__ bool genp_grid_barrier(unsigned* bar, unsigned target, int* ok_lds, int limit = GENP_SPIN_LIMIT) { genp_arrive<LOCAL>(bar); return genp_wait(bar, target, ok_lds, limit); }

// three sums over the workgroup at once: fixed tree (wave sums by DPP, then the eight in order): every workgroup forms the same values
__device__ __forceinline__ void genp_block_sum3(float& a, float& b, float& c, float* red24) {
  a = wave_sum_fast(a); b = wave_sum_fast(b); c = wave_sum_fast(c);
  const int w = threadIdx.x >> 6;
  lds_barrier();
  if ((threadIdx.x & 63) == 0) { red24[w] = a; red24[8 + w] = b; red24[16 + w] = c; }
  lds_barrier();
  a = ((red24[0] + red24[1]) + (red24[2] + red24[3])) + ((red24[4] + red24[5]) + (red24[6] + red24[7]));
  b = ((red24[8] + red24[9]) + (red24[10] + red24[11])) + ((red24[12] + red24[13]) + (red24[14] + red24[15]));
  c = ((red24[16] + red24[17]) + (red24[18] + red24[19])) + ((red24[20] + red24[21]) + (red24[22] + red24[23]));
}
__device__ __forceinline__ float genp_block_sum(float v, float* red8) {
  v = wave_sum_fast(v);
  const int w = threadIdx.x >> 6;
  lds_barrier();
  if ((threadIdx.x & 63) == 0) red8[w] = v;
  lds_barrier();
  return ((red8[0] + red8[1]) + (red8[2] + red8[3])) + ((red8[4] + red8[5]) + (red8[6] + red8[7]));
}

// the body for one store / load policy (LOCAL: every workgroup of the launch sits on the same XCD)
template <bool LOCAL>
__device__ __forceinline__ void genp_body(const GenNet& net, const GenArgs& a, const GenPersist& pp, const int wg, float* const sm, unsigned bar_n) {
  const int RS = pp.RS, G = pp.G, H = pp.H, NBR = pp.NBR;
  const bool tile_wg = wg < G * NBR;           // the other workgroups only help with the reduce / Adam phases
  const int g = tile_wg ? wg / NBR : 0, b = tile_wg ? wg - g * NBR : 0;
  float* const ACT = sm;                       // [16][RS] observation + every layer's output
  float* const DZ = sm + 16 * RS;              // [16][RS] d loss / d pre-activation
  float* const SCR = DZ + 16 * RS;             // K-split partial sums of the backward pass
  float* const G2 = SCR + GENP_SCR;            // [16][16] log_std gradient terms of the rows
  float* const RST = G2 + 256;                 // [3][16][8] per-row loss terms; the tiles' sums and the slices' norms in the Adam phase
  float* const RED = RST + 512;                // [32] reduction scratch
  float* const REC = RED + 32;                 // [16][GENP_REC] the rows' records
  float* const ADV = REC + 16 * GENP_REC;      // [16] the minibatch's advantage sums
  float* const LS = ADV + 16;                  // [16] log_std
  int* const IW = reinterpret_cast<int*>(LS + 16);
  int* const RIDX = IW;                        // [16] storage offsets of the rows
  int* const RIDXN = IW + 16;                  // [16] ... of the next minibatch's rows
  const GenpTables T = genp_tables(IW);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int n = net.n, O = net.O, A = net.A, OB = pp.OB;
  const bool discrete = net.discrete != 0;
  const __amdgpu_buffer_rsrc_t rsP = genp_rsrc(a.params, (size_t)n), rsPart = genp_rsrc(pp.part, (size_t)G * n);
  const __amdgpu_buffer_rsrc_t rsXT = genp_rsrc(pp.xt, (size_t)G * 3 * 16 * (pp.WTP > 0 ? pp.WTP : 1));
  const int lo = wg * pp.slice, hi = (lo + pp.slice < n) ? lo + pp.slice : n;      // this workgroup's parameter slice
  const int role_mask = NBR == 1 ? 7 : 1 << b;      // the heads (0 policy, 1 value, 2 cost value) whose loss this workgroup forms
  const bool role0 = (role_mask & 1) != 0;
  if (tile_wg && tid == 0) genp_build(net, pp.poff, NBR, b, pp.owner, T);
  __syncthreads();
  const int n_stages = T.MISC[3], trunk_last = T.MISC[4];
#define SU(x) __builtin_amdgcn_readfirstlane(x)      // a workgroup-uniform value read from LDS, back into a scalar register
  // replicated logged state (identical in every workgroup)
  float acc_ent = 0.f, acc_pg = 0.f, acc_cf = 0.f, acc_vr = 0.f, acc_vc = 0.f, last_pol = 0.f, last_vr = 0.f, last_vc = 0.f, kl_acc = 0.f, mean_kl = 0.f;
  int steps_done = 0, early_stop_epoch = a.hp.n_epochs;
  const float nu = a.nu[0];
  bool stop = false, aborted = false;
  // hp._pad & 64 (tests): the last workgroup leaves before the first step — the others' waits must give up (a short limit) and the launch must end with stats[11] set
  const int spin_limit = (a.hp._pad & 64) ? (1 << 16) : GENP_SPIN_LIMIT;
  if ((a.hp._pad & 64) && wg == H - 1) return;
  const bool prof = (a.hp._pad & 1) != 0 && wg == 0;       // phase timers of workgroup 0, thread 0 (tools/generic_only.py PROF=1): stats[12..21]
  unsigned long long* const PH = reinterpret_cast<unsigned long long*>(IW + GENP_INTS);      // [16] cycles per phase (in LDS: 34 scalar registers otherwise); 10..15: parts of the phases before them (stats[23..28])
  unsigned long long t_last = prof ? stamp() : 0ull;
#ifdef GENP_X_STAGES
#define GSUB(k)
#define GSTAGE(k) GSTAMP(k)
#else
#define GSUB(k) GSTAMP(k)
#define GSTAGE(k)
#endif
#define GSTAMP(k) if (prof) { const unsigned long long now_ = stamp(); if (tid == 0) PH[k] += now_ - t_last; t_last = now_; }

  // ---- the rows of a minibatch: fetched in two round trips (row offsets, then what they address) into registers, committed to LDS later
  int nx_idx = -1, nx_ridx = 0;
  const int ob_e = tid < 16 * OB ? tid : 0, ob_rr = ob_e / OB, ob_k = ob_e - ob_rr * OB;
  float nx_ar = 0.f, nx_ac = 0.f, nx_obs[4] = {0.f, 0.f, 0.f, 0.f}, nx_rec = 0.f;
  auto rows_issue1 = [&](const PlanStep& pn) {
    const int nbn = pn.nb_flags & NB_MASK;
    nx_idx = -1;
    if (nbn == 0) return;
    if (role0 && tid < nbn) nx_idx = a.perm_off[pn.perm_base + tid];
    if (tid < 16) nx_ridx = a.perm_off[pn.perm_base + (16 * g + tid < nbn ? 16 * g + tid : 0)];
  };
  // (the loads of the second trip are UNCONDITIONAL — a lane with nothing to fetch re-reads element 0 — so that their number is static and
  // the compiler can wait for the loads issued before them with vmcnt(7) instead of vmcnt(0))
  auto rows_issue2 = [&]() {
    if (tid < 16) RIDXN[tid] = nx_ridx;
    lds_barrier();
    const int ia = nx_idx >= 0 ? nx_idx : 0;
    nx_ar = a.buf.reward_advantages[ia]; nx_ac = a.buf.cost_advantages[ia];
    nx_obs[0] = a.buf.observations[(size_t)RIDXN[ob_rr] * O + (ob_k < O ? ob_k : 0)];      // (element tid of the 16 x OB block: row / column found once, before the loop)
#pragma unroll
    for (int u = 1; u < 4; ++u) {
      const int e = tid + GENP_TH * u, ec = e < 16 * OB ? e : 0;
      const int rr = ec / OB, k = ec - rr * OB;
      nx_obs[u] = a.buf.observations[(size_t)RIDXN[rr] * O + (k < O ? k : 0)];
    }
    {
      const int tc = tid < 16 * GENP_REC ? tid : 0;
      const int rr = tc / GENP_REC, f = tc - rr * GENP_REC, idx = RIDXN[rr];
      const float* ptr = a.buf.actions + (size_t)idx * a.buf.act_store + (f < a.buf.act_store ? f : 0);
      if (f == 16) ptr = a.buf.log_probs + idx;
      if (f == 17) ptr = a.buf.reward_advantages + idx;
      if (f == 18) ptr = a.buf.cost_advantages + idx;
      if (f == 19) ptr = a.buf.reward_returns + idx;
      if (f == 20) ptr = a.buf.cost_returns + idx;
      if (f == 21) ptr = a.buf.reward_values + idx;
      if (f == 22) ptr = a.buf.cost_values + idx;
      nx_rec = *ptr;
    }
  };
  auto rows_commit = [&](const PlanStep& pn) {
    const int nbn = pn.nb_flags & NB_MASK;
    if (nbn == 0) return;
    if (tid < 16 * OB) ACT[ob_rr * RS + ob_k] = ob_k < O ? nx_obs[0] : 0.f;
    if (16 * OB > GENP_TH) {
#pragma unroll
      for (int u = 1; u < 4; ++u) {
        const int e = tid + GENP_TH * u;
        if (e < 16 * OB) { const int rr = e / OB, k = e - rr * OB; ACT[rr * RS + k] = k < O ? nx_obs[u] : 0.f; }
      }
    }
    for (int e = tid + 4 * GENP_TH; e < 16 * OB; e += GENP_TH) {      // (observations wider than 128: the rest, not prefetched)
      const int rr = e / OB, k = e - rr * OB;
      ACT[rr * RS + k] = k < O ? a.buf.observations[(size_t)RIDXN[rr] * O + k] : 0.f;
    }
    if (tid < 16 * GENP_REC) {
      const int f = tid % GENP_REC;
      REC[tid] = (f < a.buf.act_store || (f >= 16 && f < 23)) ? nx_rec : 0.f;
    }
    if (tid < 16) RIDX[tid] = RIDXN[tid];
    float sr = nx_idx >= 0 ? nx_ar : 0.f, sc = nx_idx >= 0 ? nx_ac : 0.f, srr = sr * sr;
    genp_block_sum3(sr, sc, srr, RED);
    if (tid == 0) { ADV[0] = sr; ADV[1] = sc; ADV[2] = srr; }
    lds_barrier();
  };
  if (tile_wg) {
    const PlanStep p0 = pp.plan[0];
    rows_issue1(p0); rows_issue2(); rows_commit(p0);
  }

  for (int st = 0; st < pp.n_steps && !stop; ++st) {
    const PlanStep ps = pp.plan[st], pn = pp.plan[st + 1];
    const int nb = ps.nb_flags & NB_MASK, epoch = ps.nb_flags >> NB_EPOCH;
    const bool first_mb = (ps.nb_flags >> NB_FIRST) & 1, last_mb = (ps.nb_flags >> NB_LAST) & 1;
    const float inv_nb = 1.f / (float)nb;
    if (tile_wg) {
    float ls_reg = 0.f;
    if (!discrete && role0 && tid < A) ls_reg = genp_ld<LOCAL>(rsP, 4u * (unsigned)(net.log_std + tid));
    const float sr = ADV[0], sc = ADV[1], srr = ADV[2];
    const float mean_r = sr * inv_nb, mean_c = sc * inv_nb;
    const float istd_r = 1.f / (sqrtf(fmaxf(srr - sr * mean_r, 0.f) / (float)(nb - 1)) + 1e-8f);
    GSTAMP(0)   // rows + advantage statistics
    f32x4 X0[8];
    // ================= forward: the wave's tiles in list order, the next tile's weights in flight under this tile's MFMAs =================
    {
      const int fn = SU(T.FLN[w]);
      const int* fl = T.FL + w * GENP_FLIST;
      struct FD { int it, in_dim, out_dim, w_off, b_off, in_col, out_col, tanh; };
      auto f_decode = [&](int idx) -> FD {
        FD d;
        d.it = SU(fl[idx]);
        const int* row = T.LT + 8 * (d.it & 31);
        d.in_dim = SU(row[0]); d.out_dim = SU(row[1]); d.w_off = SU(row[2]); d.b_off = SU(row[3]); d.in_col = SU(row[4]); d.out_col = SU(row[5]); d.tanh = SU(row[6]) & 1;
        return d;
      };
      // ONE register image of a tile's weights: chunk c of the NEXT item is loaded into X[c] right after this item's MFMAs have read
      // it — every load is issued about one item ahead of its use, also across the stage barriers (the weights do not change inside a step)
      f32x4 Bv = f32x4{0.f, 0.f, 0.f, 0.f}, acc = Bv;
      // (every load of the loop is UNCONDITIONAL — an absent chunk reads at an offset outside the resource — so that the number of loads
      // in flight is the same on every path and the compiler waits with vmcnt(8) instead of vmcnt(0): a load under a branch made it
      // wait for the prefetch it had just issued, a full L2 round trip per chunk)
      auto f_row = [&](const FD& d) -> unsigned {      // byte offset of this lane's 4 weights of chunk 0
        const int t = (d.it >> 5) & 15, kb = (d.it >> 9) & 7, m = 16 * t + r;
        return (d.it >= 0 && m < d.out_dim) ? 4u * (unsigned)(d.w_off + m * d.in_dim + 128 * kb + 4 * q) : GENP_OOB;
      };
      auto f_bias = [&](const FD& d) -> f32x4 {      // (only the first K slab of a tile starts from the bias)
        const int t = (d.it >> 5) & 15;
        return genp_ld4<LOCAL>(rsP, (d.it >= 0 && ((d.it >> 9) & 7) == 0 && 16 * t + 4 * q < d.out_dim) ? 4u * (unsigned)(d.b_off + 16 * t + 4 * q) : GENP_OOB);
      };
      int fi = 0;
      FD d, nd;
      d.it = -1; d.in_dim = 0; d.out_dim = 0; d.w_off = 0; d.b_off = 0; d.in_col = 0; d.out_col = 0; d.tanh = 0;
      nd = d;
      if (fn > 0) d = f_decode(0);
      {
        const unsigned base = f_row(d);
        const int kk = d.it >= 0 ? d.in_dim - 128 * ((d.it >> 9) & 7) : 0;
        Bv = f_bias(d);
#pragma unroll
        for (int c = 0; c < 8; ++c) X0[c] = genp_ld4<LOCAL>(rsP, 16 * c < kk ? base + 64u * (unsigned)c : GENP_OOB);
      }
      for (int s = 0; s < n_stages; ++s) {
        while (fi < fn && (d.it >> 13) == s) {
          const bool more = fi + 1 < fn;
          nd.it = -1;
          if (more) nd = f_decode(fi + 1);
          const unsigned nbase = f_row(nd);
          const int nkk = more ? nd.in_dim - 128 * ((nd.it >> 9) & 7) : 0;
          const int t = (d.it >> 5) & 15, kb = (d.it >> 9) & 7, lastk = (d.it >> 12) & 1, kk = d.in_dim - 128 * kb;
          if (kb == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = 16 * t + 4 * q + i < d.out_dim ? Bv[i] : 0.f;
          }
          Bv = f_bias(nd);
          const float* brow = ACT + r * RS + d.in_col + 128 * kb + 4 * q;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (16 * c < kk) {
              const f32x4 bv = lds128(brow + 16 * c);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc = MFMA_F32(X0[c][e], bv[e], acc);
            }
            X0[c] = genp_ld4<LOCAL>(rsP, 16 * c < nkk ? nbase + 64u * (unsigned)c : GENP_OOB);
          }
          if (lastk) {
            if (d.tanh) {
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[i] = fast_tanh(acc[i]);
            }
            *reinterpret_cast<f32x4*>(ACT + r * RS + d.out_col + 16 * t + 4 * q) = acc;
          }
          d = nd; ++fi;
          if (fi >= fn) d.it = -1;
        }
        lds_barrier();
        if (s < 3) { GSTAGE(10 + s) }
      }
    }
    if (tid < A) LS[tid] = ls_reg;
    lds_barrier();
    GSTAMP(1)   // forward
    // ================= loss terms of the heads this workgroup owns =================
    // policy: waves 0..3, lane (o = lane % 16, row = 4 w + lane / 16) — the sums over the outputs by DPP over the 16 lanes of a row;
    // value / cost value: waves 4 / 5, lane = row
    if (w < 4 && role0) {
      const int o = r, row = 4 * w + q;
      const bool valid = 16 * g + row < nb, on = o < A;
      const float* rec = REC + row * GENP_REC;
      const int hcol = T.MISC[0];
      const float out = ACT[row * RS + hcol + o];      // (pad columns of the head are zeros)
      const float old_lp = rec[16];
      float lp, ent, g1, g2;      // log-prob and entropy of the row (in every lane), this output's d lp / d mean (or logit), d lp / d log_std
      if (!discrete) {
        const float ls = on ? LS[o] : 0.f, dd = rec[o & 15] - out, sd = __expf(ls), iv = 1.f / (sd * sd);
        lp = row_sum(on ? -(dd * dd) * (0.5f * iv) - ls - LOG_SQRT_2PI_F : 0.f);
        ent = row_sum(on ? HALF_LOG_2PI_PLUS_HALF_F + ls : 0.f);
        g1 = dd * iv; g2 = (dd * dd) * iv - 1.f;
      } else {
        const float m = row_max(on ? out : -INFINITY);
        const float lse = m + logf(row_sum(on ? expf(out - m) : 0.f));
        const int action = (int)rec[0];
        const float lg = out - lse, pr = expf(lg);
        ent = -row_sum(on ? pr * lg : 0.f);
        lp = row_sum(o == action ? lg : 0.f);
        g1 = (o == action ? 1.f : 0.f) - pr; g2 = pr * (lg + ent);
      }
      const float ratio = __expf(lp - old_lp);
      const float Ar = (rec[17] - mean_r) * istd_r;
      const float Ac = rec[18] - mean_c;
      const float clip = a.hp.clip_range;
      const float s1 = Ar * ratio, s2 = Ar * fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
      const float gsel = (s1 <= s2) ? Ar : 0.f;
      const float dlp = valid ? inv_nb / (1.f + nu) * (-gsel + nu * Ac) * ratio : 0.f;
      const float dent = valid ? a.hp.ent_coef * inv_nb : 0.f;
      if (on) DZ[row * RS + hcol + o] = discrete ? dlp * g1 + dent * g2 : dlp * g1;
      G2[row * 16 + o] = (on && !discrete) ? dlp * g2 : 0.f;
      if (o == 0) {
        float* rs = RST + row * 8;
        rs[0] = valid ? fminf(s1, s2) : 0.f; rs[1] = valid ? Ac * ratio : 0.f; rs[2] = (valid && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
        rs[3] = valid ? old_lp - lp : 0.f; rs[4] = valid ? ent : 0.f;
      }
    } else if ((w == 4 || w == 5) && lane < 16 && ((role_mask >> (w - 3)) & 1)) {
      const int role = w - 3, row = lane;
      const bool valid = 16 * g + row < nb;
      const float* rec = REC + row * GENP_REC;
      const int hcol = T.MISC[role];
      const float v = ACT[row * RS + hcol];
      const float R = role == 1 ? rec[19] : rec[20];
      const float vclip = role == 1 ? a.hp.clip_range_reward_vf : a.hp.clip_range_cost_vf;
      const float vcoef = role == 1 ? a.hp.reward_vf_coef : a.hp.cost_vf_coef;
      float vp = v, pass = 1.f;
      if (vclip >= 0.f) {
        const float old = role == 1 ? rec[21] : rec[22];
        const float dv = v - old;
        vp = old + fminf(fmaxf(dv, -vclip), vclip);
        pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
      }
      const float e = vp - R;
      DZ[row * RS + hcol] = valid ? vcoef * 2.f * e * inv_nb * pass : 0.f;
      RST[(role * 16 + row) * 8] = valid ? e * e : 0.f;
    }
    lds_barrier();
    GSTAMP(2)   // loss
    // ================= backward of the activations, stage by stage: d h = sum over the layers that read h of W^T dz (layer order) =================
    {
      const int bn = SU(T.BLN[w]);
      const int* bl = T.BL + w * GENP_BLIST;
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      struct BD { int it, nl, c_out, c_woff, c_col; };      // the item, the width of l (= the consumer's row stride), the consumer's units / weights / d z column
      auto b_decode = [&](int idx) -> BD {
        BD d;
        d.it = SU(bl[idx]);
        const int* row = T.LT + 8 * (d.it & 31);
        d.nl = SU(row[0]); d.c_out = SU(row[1]); d.c_woff = SU(row[2]); d.c_col = SU(row[5]);
        return d;
      };
      auto b_load = [&](const BD& d, int u, int e) -> f32x4 {      // the consumer's weights [unit j0 + 16 u + 4 q + e][64 T + 4 r ..]; an absent chunk: outside the resource
        const int TT = (d.it >> 15) & 3, j = 16 * ((d.it >> 5) & 15) + 16 * u + 4 * q + e;
        const bool there = d.it >= 0 && !((d.it >> 25) & 1) && u <= ((d.it >> 9) & 1);
        return genp_ld4<LOCAL>(rsP, (there && j < d.c_out) ? 4u * (unsigned)(d.c_woff + j * d.nl + 64 * TT + 4 * r) : GENP_OOB);
      };
      auto b_chunks = [&](const BD& d) -> int { return ((d.it >> 25) & 1) ? 0 : 1 + ((d.it >> 9) & 1); };
      auto b_flush = [&](const BD& d) {      // acc[i][ii] = d h[row 4 q + ii][unit 64 T + 4 r + i] of this part
        const int l = (d.it >> 10) & 31, TT = (d.it >> 15) & 3, p = (d.it >> 17) & 7, s = (d.it >> 20) & 15;
        const int* row = T.LT + 8 * l;
        const int nl = SU(row[1]), lcol = SU(row[5]), scol = SU(row[7]), k = 64 * TT + 4 * r;
        if (SU(T.SP[s]) == 0) {      // straight into the image
          if (k < (nl + 15) / 16 * 16) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
              const int rw = 4 * q + ii;
              const f32x4 h = lds128(ACT + rw * RS + lcol + k);
              f32x4 v;
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = k + i < nl ? fmaf(-(h[i] * h[i]), acc[i][ii], acc[i][ii]) : 0.f;
              *reinterpret_cast<f32x4*>(DZ + rw * RS + lcol + k) = v;
            }
          }
        } else {
          const int SW = SU(T.SWD[s]);
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
            *reinterpret_cast<f32x4*>(SCR + (p * 16 + 4 * q + ii) * SW + scol + k) = f32x4{acc[0][ii], acc[1][ii], acc[2][ii], acc[3][ii]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      };
      int bi = 0;
      BD d, nd;
      d.it = -1; d.nl = 0; d.c_out = 0; d.c_woff = 0; d.c_col = 0;
      nd = d;
      if (bn > 0) d = b_decode(0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) X0[4 * u + e] = b_load(d, u, e);
      for (int s = n_stages - 2; s >= 0; --s) {
        const int P = SU(T.SP[s]), SW = SU(T.SWD[s]);
        if (SW == 0) continue;
        while (bi < bn && ((d.it >> 20) & 15) == s) {
          nd.it = -1;
          if (bi + 1 < bn) nd = b_decode(bi + 1);
          const int nch = b_chunks(d), j0 = 16 * ((d.it >> 5) & 15);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u < nch) {
              const f32x4 dzv = lds128(DZ + r * RS + d.c_col + j0 + 16 * u + 4 * q);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = MFMA_F32(dzv[e], X0[4 * u + e][i], acc[i]);
            }
            // (the next item's weights, into the registers this chunk has just been read from; unconditional: see the forward pass)
#pragma unroll
            for (int e = 0; e < 4; ++e) X0[4 * u + e] = b_load(nd, u, e);
          }
          if ((d.it >> 24) & 1) b_flush(d);
          d = nd; ++bi;
          if (bi >= bn) d.it = -1;
        }
        lds_barrier();
        if (P != 0) {      // the parts' sums: x (1 - h^2) into the image, or (a trunk under three branch workgroups) this branch's share to memory
          for (int si = 0; si < 3; ++si) {
            const int l = SU(T.STL[4 * s + si]);
            if (l < 0) break;
            const int* row = T.LT + 8 * l;
            const int nl = SU(row[1]), lcol = SU(row[5]), role = (SU(row[6]) >> 1) & 3, scol = SU(row[7]), np4 = (nl + 15) / 16 * 4;      // quads of a padded row
            for (int idx = tid; idx < 16 * np4; idx += GENP_TH) {
              const int rw = idx / np4, k = 4 * (idx - rw * np4);
              f32x4 v = lds128(SCR + rw * SW + scol + k);
              for (int p = 1; p < P; ++p) {
                const f32x4 x = lds128(SCR + (p * 16 + rw) * SW + scol + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += x[i];
              }
              if (role == 1) {
                const f32x4 h = lds128(ACT + rw * RS + lcol + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = k + i < nl ? fmaf(-(h[i] * h[i]), v[i], v[i]) : 0.f;
                *reinterpret_cast<f32x4*>(DZ + rw * RS + lcol + k) = v;
              } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = k + i < nl ? v[i] : 0.f;
                genp_st4<LOCAL>(rsXT, 4u * (unsigned)(((g * 3 + b) * 16 + rw) * pp.WTP + k), v);
              }
            }
          }
          if (NBR == 3 && trunk_last >= 0 && SU(T.STL[4 * s]) == trunk_last) {      // the three shares of d h(trunk) meet
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
              if (LOCAL) __hip_atomic_store(pp.xflag + g * 3 + b, (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              else __hip_atomic_store(pp.xflag + g * 3 + b, (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int* row = T.LT + 8 * trunk_last;
            if (((SU(row[6]) >> 1) & 3) == 3) {
              if (tid < 3) {      // bounded like genp_wait: the launch's spin limit, and the abort word of a workgroup that already gave up
                int spins = 0;
                while (__hip_atomic_load(pp.xflag + g * 3 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(st + 1)) {
                  if (++spins > spin_limit || ((spins & 1023) == 0 && __hip_atomic_load(pp.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store(pp.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next barrier ends the launch)
                    break;
                  }
                  __builtin_amdgcn_s_sleep(1);
                }
              }
              __syncthreads();
                          const int nl = SU(row[1]), lcol = SU(row[5]), WT = pp.WTP;
              for (int idx = tid; idx < 4 * WT; idx += GENP_TH) {
                const int rw = idx / (WT / 4), k = 4 * (idx - rw * (WT / 4));
                const f32x4 x0 = genp_ld4<LOCAL>(rsXT, 4u * (unsigned)(((g * 3 + 0) * 16 + rw) * WT + k));
                const f32x4 x1 = genp_ld4<LOCAL>(rsXT, 4u * (unsigned)(((g * 3 + 1) * 16 + rw) * WT + k));
                const f32x4 x2 = genp_ld4<LOCAL>(rsXT, 4u * (unsigned)(((g * 3 + 2) * 16 + rw) * WT + k));
                const f32x4 h = lds128(ACT + rw * RS + lcol + k);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float sum = (x0[i] + x1[i]) + x2[i]; v[i] = k + i < nl ? fmaf(-(h[i] * h[i]), sum, sum) : 0.f; }
                *reinterpret_cast<f32x4*>(DZ + rw * RS + lcol + k) = v;
              }
            }
          }
          lds_barrier();
        }
      }
    }
    GSTAMP(3)   // backward
    // ================= partial weight gradients of this tile's 16 rows -> part[g] =================
    float* const mypart = pp.part + (size_t)g * n;
    {
      // item = 16 units (jt) of a layer, dealt round-robin to the waves; inside, the layer's inputs 64 at a time (four 16 x 16 tiles, K = the
      // 16 rows): d W^T tile = A (inputs x rows, from the ACT image) x B (rows x units, from the DZ image) -> lane (unit r, q) holds inputs
      // 4 q .. 4 q + 3 of its unit: ONE 16-byte store per tile.  The per-item work outside the MFMAs is kept to pointer increments (the
      // kernel spends 4 cycles per vector instruction and nothing overlaps an MFMA of the same SIMD).
      const int lane_act = q * RS + r;      // row q (+ 4 e), column r of a tile
      int base = 0;
      for (int wi = 0; wi < GEN_MAX_LAYERS; ++wi) {
        const int l = SU(T.WL[wi]);
        if (l < 0) break;
        const int* row = T.LT + 8 * l;
        const int in_dim = SU(row[0]), out_dim = SU(row[1]), w_off = SU(row[2]), b_off = SU(row[3]), in_col = SU(row[4]), out_col = SU(row[5]);
        const int nj = (out_dim + 15) / 16, nkt = (in_dim + 15) / 16, nfull = (in_dim & 3) == 0 ? nkt : in_dim / 16;      // tiles a 16-byte store serves (the last one too when rows are whole quads)
        for (int jt = (w - base) & 7; jt < nj; jt += 8) {
          float az[4];      // d z[row 4 e + q][unit 16 jt + r]
#pragma unroll
          for (int e = 0; e < 4; ++e) az[e] = DZ[lane_act + 4 * e * RS + out_col + 16 * jt];
          const int j = 16 * jt + r;
          const unsigned jrow = j < out_dim ? 4u * (unsigned)(g * n + w_off + j * in_dim + 4 * q) : GENP_OOB / 2;      // this lane's 16 bytes of tile 0 (outside the layer: dropped)
          const float* ap = ACT + lane_act + in_col;
          for (int k0 = 0; k0 < nkt; k0 += 4) {
            float ax[4][4];   // the layer's input [row 4 e + q][16 (k0 + u) + r]: all sixteen reads ahead of the MFMAs (an absent tile re-reads the last one)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int kt = k0 + u < nkt ? k0 + u : nkt - 1;
#pragma unroll
              for (int e = 0; e < 4; ++e) ax[u][e] = ap[4 * e * RS + 16 * kt];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (k0 + u < nkt) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = MFMA_F32(ax[u][e], az[e], acc);
                const int kt = k0 + u;
                if (kt < nfull) {      // (a quad is inside the row or outside it as a whole)
                  const unsigned vo = 16 * kt + 4 * q < in_dim ? jrow : GENP_OOB / 2;
                  if (LOCAL) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rsPart, (int)vo, 64 * kt, 1);
                  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rsPart, (int)vo, 64 * kt, 16);
                } else {
#pragma unroll
                  for (int i = 0; i < 4; ++i) genp_st1<LOCAL>(rsPart, (16 * kt + 4 * q + i < in_dim ? jrow : GENP_OOB / 2) + 4u * (unsigned)(16 * kt + i), acc[i]);
                }
              }
            }
          }
          // bias: the sum over the 16 rows (lanes r = unit, q, e = row 4 e + q)
          const float sb = quad_rows_sum((az[0] + az[1]) + (az[2] + az[3]));
          if (q == 0 && j < out_dim) st_x<LOCAL>(mypart + b_off + j, sb);
        }
        base += nj;
        if (wi == 0) { GSTAGE(13) }
      }
      GSTAGE(14)
      if (w == 6 && !discrete && role0) {      // log_std: sum over the 16 rows — lane (o, q): rows 4 q .. 4 q + 3, then over q
        const float sl = quad_rows_sum((G2[(4 * q) * 16 + r] + G2[(4 * q + 1) * 16 + r]) + (G2[(4 * q + 2) * 16 + r] + G2[(4 * q + 3) * 16 + r]));
        if (q == 0 && r < A) st_x<LOCAL>(mypart + net.log_std + r, sl);
      }
      if (w == 7) {      // this tile's loss sums: policy terms 0..4, reward / cost value errors 5, 6 — lane (row, q): terms q and q + 4
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int k = q + 4 * h2, role = k < 5 ? 0 : k - 4;
          const float v = row_sum(k < 7 ? RST[(role * 16 + r) * 8 + (k < 5 ? k : 0)] : 0.f);
          if (r == 0 && k < 7 && ((role_mask >> role) & 1)) st_x<LOCAL>(pp.stat + g * 8 + k, v);
        }
      }
    }
    }      // tile_wg
    GSTAMP(4)   // weight gradients
    bar_n += H; genp_arrive<LOCAL>(pp.bar);
    GSUB(14)  // (arrival at A)
    if (tile_wg) rows_issue1(pn);      // (the next minibatch's rows, under the barriers: first round trip — the row offsets)
    if (!genp_wait(pp.bar, bar_n, T.MISC + 15, spin_limit)) { aborted = true; break; }          // (A) every tile's partials are in memory
    GSTAMP(5)   // barrier A
    // ================= this workgroup's parameter slice: sum over the tiles in tile order, squared norm =================
    // (elements lo + 4 tid .. + 3 stay in registers until Adam; a slice above 2 048 parameters walks the rest through `grad`)
    float ss = 0.f;
    const int e0 = lo + 4 * tid;
    const unsigned eo = e0 < hi ? 4u * (unsigned)e0 : GENP_OOB;
    f32x4 g0 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto slice_sum = [&](int eb) -> f32x4 {      // four consecutive elements: G 16-byte loads in flight, summed in tile order (absent tiles read 0)
      f32x4 gs = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < G; t += 4) {
        f32x4 pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = genp_ld4<LOCAL>(rsPart, (t + u < G && eb < hi) ? 4u * (unsigned)((t + u) * n + eb) : GENP_OOB);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) gs[i] += pv[u][i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = eb + i;
        if (!discrete && e >= net.log_std && e < net.log_std + A) gs[i] += -a.hp.ent_coef;      // d(ent_coef * -mean H) / d log_std
        if (e >= hi) gs[i] = 0.f;
        ss = fmaf(gs[i], gs[i], ss);
      }
      return gs;
    };
    g0 = slice_sum(e0);
    for (int e1 = e0 + 4 * GENP_TH; e1 < hi; e1 += 4 * GENP_TH) {
      const f32x4 gs = slice_sum(e1);
#pragma unroll
      for (int i = 0; i < 4; ++i) if (e1 + i < hi) pp.grad[e1 + i] = gs[i];
    }
    ss = genp_block_sum(ss, RED);
    if (tid == 0) st_x<LOCAL>(pp.norm + wg, ss);
    GSTAMP(6)   // reduce + norm
    bar_n += H; genp_arrive<LOCAL>(pp.bar);
    GSUB(10)  // (arrival at B)
    if (!genp_wait(pp.bar, bar_n, T.MISC + 15, spin_limit)) { aborted = true; break; }          // (B) every slice's squared norm is in memory
    GSTAMP(7)   // barrier B
    float total = 0.f, q7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {   // every tile's loss sums and every slice's squared norm fetched side by side (one value per thread), summed by fixed trees;
        // Adam's operands of the slice travel with them, and BEHIND them the second round trip of the next minibatch's rows (loads come back in order)
      const __amdgpu_buffer_rsrc_t rsX = genp_rsrc(pp.stat, 512);      // [stat G x 8 | norm H]
      const __amdgpu_buffer_rsrc_t rsM = genp_rsrc(a.exp_avg, (size_t)n), rsV = genp_rsrc(a.exp_avg_sq, (size_t)n);
      const float sv = genp_ld<LOCAL>(rsX, (tid < 8 * G || (tid >= 256 && tid < 256 + H)) ? 4u * (unsigned)tid : GENP_OOB);
      const f32x4 pw = genp_ld4<LOCAL>(rsP, eo), mv = genp_ld4<LOCAL>(rsM, eo), vv = genp_ld4<LOCAL>(rsV, eo);
      if (tile_wg) rows_issue2();
      const float b1 = a.hp.adam_beta1, b2 = a.hp.adam_beta2, w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
      lds_barrier();
      RST[tid] = sv;
      lds_barrier();
      total = wave_sum_fast(lane < H ? RST[256 + lane] : 0.f);
#pragma unroll
      for (int k = 0; k < 7; ++k) q7[k] = wave_sum_fast(lane < G ? RST[lane * 8 + k] : 0.f);
      GSUB(11)  // (the sums and the slice's operands are here)
      const float cc = a.hp.max_grad_norm / (sqrtf(total) + 1e-6f), coef = cc > 1.f ? 1.f : cc;
      {
        f32x4 nm, nv, nw;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float gc = g0[u] * coef;
          nm[u] = fmaf(w1, gc, b1 * mv[u]);
          nv[u] = fmaf(w2, gc * gc, b2 * vv[u]);
          nw[u] = fmaf(-ps.step_size, nm[u] / fmaf(sqrtf(nv[u]), ps.inv_bc2_sqrt, a.hp.adam_eps), pw[u]);
        }
        genp_st4<LOCAL>(rsM, eo, nm); genp_st4<LOCAL>(rsV, eo, nv); genp_st4<LOCAL>(rsP, eo, nw);      // (dwords beyond n are dropped)
      }
      for (int e1 = e0 + 4 * GENP_TH; e1 < hi; e1 += 4 * GENP_TH) {      // (slices above 2 048 parameters: the rest)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e1 + u;
          if (e < hi) {
            const float gc = pp.grad[e] * coef;
            const float m = fmaf(w1, gc, b1 * a.exp_avg[e]);
            const float v = fmaf(w2, gc * gc, b2 * a.exp_avg_sq[e]);
            a.exp_avg[e] = m; a.exp_avg_sq[e] = v;
            st_x<LOCAL>(a.params + e, fmaf(-ps.step_size, m / fmaf(sqrtf(v), ps.inv_bc2_sqrt, a.hp.adam_eps), genp_ld<LOCAL>(rsP, 4u * (unsigned)e)));
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
        if (wg == 0 && tid == 0) a.stats[32 + epoch] = mean_kl;
        if (a.hp.use_target_kl && mean_kl > 1.5f * a.hp.target_kl) { stop = true; early_stop_epoch = epoch; }
      }
    }
    GSTAMP(8)   // Adam + logged sums
    bar_n += H; genp_arrive<LOCAL>(pp.bar);
    GSUB(12)  // (arrival at C)
    if (tile_wg) rows_commit(pn);      // (into the image)
    GSUB(13)  // (the rows in the image)
    if (!genp_wait(pp.bar, bar_n, T.MISC + 15, spin_limit)) { aborted = true; break; }          // (C) the updated parameters are in memory
    GSTAMP(9)   // barrier C
  }
  if (prof && tid == 0) a.stats[22] = LOCAL ? 1.f : 0.f;
  if (prof && tid == 0)
    for (int k = 0; k < 16; ++k) a.stats[(k < 10 ? 12 : 13) + k] = (float)((double)PH[k] / (double)(steps_done > 0 ? steps_done : 1));
  if (wg == 0 && tid == 0) {
    a.stats[0] = (float)early_stop_epoch;
    a.stats[1] = (float)steps_done;
    a.stats[2] = acc_ent; a.stats[3] = acc_pg; a.stats[4] = acc_vr; a.stats[5] = acc_vc; a.stats[6] = acc_cf;
    a.stats[7] = mean_kl; a.stats[8] = last_pol; a.stats[9] = last_vr; a.stats[10] = last_vc; a.stats[11] = aborted ? 1.f : 0.f;
    const_cast<int*>(a.adam_t)[0] += steps_done;
  }
}

__global__ void __launch_bounds__(GENP_TH) gen_train_persistent_kernel(GenNet net, GenArgs a, GenPersist pp, int packed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // packed: a 1-D grid of 8 (H - 1) + 1 workgroups of which every eighth works — workgroups are dealt round-robin over the 8 XCDs, so
  // the H working ones land on ONE (ppo_common.h: XCD placement); the others leave at once
  if (packed && (blockIdx.x & (XCD_STRIDE - 1)) != 0) return;
  const int H = pp.H, wg = packed ? (int)blockIdx.x / XCD_STRIDE : (int)blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 32 * pp.RS + GENP_FLOATS; i += GENP_TH) sm[i] = 0.f;
  // (agent scope: a plain store would sit dirty in THIS XCD's L2, and on a grid spread over several XCDs the owner's polls — served by its own L2 for
  // lines dirty there — would read that zero for ever: `ICRL_NO_XCD_PACK=1` with a shared trunk hung until this was a write-through store)
  if (tid < 3 && wg < pp.G) __hip_atomic_store(pp.xflag + wg * 3 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  // do all workgroups share an XCD?  (every workgroup publishes its XCC id in its norm slot, one barrier, everybody compares)
  unsigned bar_n = 0;
  if (tid == 0) __hip_atomic_store(pp.norm + wg, __uint_as_float(0x100u | xcc_id()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int* const ok_lds = reinterpret_cast<int*>(sm + 32 * pp.RS + GENP_FLOATS - 33);      // (an int of the table region the body does not use before its own first barrier: MISC is rebuilt there)
  bar_n += H;
  const bool ok1 = genp_grid_barrier<false>(pp.bar, bar_n, ok_lds);
  bool local = true;
  for (int t = 0; t < H; ++t) local = local && __float_as_uint(ld_sc1(pp.norm + t)) == (0x100u | xcc_id());
  local = __builtin_amdgcn_readfirstlane((int)local) != 0;
  bar_n += H;
  const bool ok2 = genp_grid_barrier<false>(pp.bar, bar_n, ok_lds);      // (the slots are reused by the first step's norms)
  if (!(ok1 && ok2)) { if (wg == 0 && tid == 0) { a.stats[0] = (float)a.hp.n_epochs; a.stats[1] = 0.f; a.stats[11] = 1.f; } return; }
  if (local) genp_body<true>(net, a, pp, wg, sm, bar_n);
  else genp_body<false>(net, a, pp, wg, sm, bar_n);
}

static_assert(ICRL_PPO_GENERIC_BYTES(64, 776, 1000) == 4 * (64 + 64 * (24 + 1 + 16 + 2 * 776) + 1000 + 4 + 1088 + 5 * 1000 + 1024), "ICRL_PPO_GENERIC_BYTES");
static_assert(ICRL_PPO_GENERIC_BYTES(1024, 776, 1000) == 4 * (64 + 1024 * (24 + 1 + 16 + 2 * 776) + 1000 + 4 + 1088), "ICRL_PPO_GENERIC_BYTES");

// the schedule table of ppo_train.hip (per optimiser step: Adam's bias corrections in double, rows, flags, permutation base)
__global__ void ppo_plan_kernel(const int* adam_t, int n_steps, int n_mb, int n_total, int B, double lr, double b1, double b2, PlanStep* steps,
                                PlanChunk* chunks, int two_per_step);

// the persistent form, when the shape allows it: 0 launched | < 0 not eligible (the caller uses the launch-per-phase form) | > 0 error
static int launch_train_generic_persistent(const GenNet& net, GenArgs& a, const icrl_ppo_hyper_t* hp, int32_t* adam_step, void* sync_ws, hipStream_t s) {
  static const bool off = getenv("ICRL_GEN_LAUNCHES") != nullptr;      // A/B and tests: the three-launches-per-step form
  static const char* nbr_env = getenv("ICRL_GEN_BRANCH_WGS");      // A/B: "1" keeps one workgroup per row tile
  const int B = hp->batch_size, G = (B + 15) / 16, n = net.n;
  if (off || G > GENP_MAX_TILES || n > GENP_MAX_PARAMS || B > NB_MASK) return -1;
  GenPersist pp;
  pp.OB = (net.O + 15) / 16 * 16;
  const long long n_steps = (long long)hp->n_epochs * a.n_mb;
  if (n_steps >= (1ll << 21)) return -1;
  // three workgroups per row tile (one per branch) while they fit one XCD; a trunk then needs room for the exchanged shares of its gradient
  pp.WTP = net.n_shared > 0 ? (net.layer[net.n_shared - 1].out_dim + 15) / 16 * 16 : 0;
  pp.NBR = (3 * G <= GENP_MAX_WGS && !(nbr_env && nbr_env[0] == '1') && (size_t)G * 48 * pp.WTP <= (size_t)n) ? 3 : 1;
  {   // the trunk's owner: the branch with the fewest parameters of its own
    int cnt[3] = {0, 0, 0};
    for (int l = net.n_shared; l < net.n_layers; ++l) cnt[net.layer[l].slot] += net.layer[l].in_dim * net.layer[l].out_dim;
    pp.owner = cnt[1] <= cnt[0] && cnt[1] <= cnt[2] ? 1 : (cnt[2] <= cnt[0] ? 2 : 0);
  }
  int off_ = pp.OB;
  for (int l = 0; l < net.n_layers; ++l) { pp.poff[l] = off_; off_ += (net.layer[l].out_dim + 15) / 16 * 16; }
  pp.RS = off_ + ((8 - off_ % 32) + 32) % 32;      // = 8 mod 32: conflict-free ds_read_b128 of a row image
  if (pp.RS > GENP_MAX_RS) return -1;
  {   // do the work lists fit?
    int ibuf[GENP_INTS];
    const GenpTables T = genp_tables(ibuf);
    for (int b = 0; b < pp.NBR; ++b)
      if (!genp_build(net, pp.poff, pp.NBR, b, pp.owner, T)) return -1;
  }
  // helpers for the reduce / Adam phases: ~2048 parameters per workgroup, all workgroups on one XCD (<= its CUs)
  int H = (n + 2047) / 2048;
  const int tiles = G * pp.NBR;
  H = H < tiles ? tiles : (H > GENP_MAX_WGS ? (tiles > GENP_MAX_WGS ? tiles : GENP_MAX_WGS) : H);
  pp.G = G; pp.H = H; pp.slice = ((n + H - 1) / H + 3) / 4 * 4;
  PlanStep* steps = reinterpret_cast<PlanStep*>((char*)sync_ws + 512);
  hipLaunchKernelGGL(ppo_plan_kernel, dim3((unsigned)((n_steps + 2 + 255) / 256)), dim3(256), 0, s, adam_step, (int)n_steps, a.n_mb, a.n_total, B,
                     (double)hp->lr, (double)hp->adam_beta1, (double)hp->adam_beta2, steps, (PlanChunk*)nullptr, 0);
  pp.plan = steps; pp.n_steps = (int)n_steps;
  float* extra = a.scratch + gen_floats(B, net.row_floats, n);      // behind the launch-per-phase layout: [stat 256 | norm 256 | flags 512 | trunk shares n | part G x n]
  pp.stat = extra; pp.norm = extra + 256; pp.xflag = reinterpret_cast<unsigned*>(extra + 512); pp.xt = extra + 1024; pp.part = extra + 1024 + n;
  pp.grad = a.scratch + gen_off_grad(B, net.row_floats);
  pp.bar = reinterpret_cast<unsigned*>(a.scratch + 60);           // (the first 64 floats are zeroed by the caller)
  const size_t lds = (size_t)(32 * pp.RS + GENP_FLOATS) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)gen_train_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  GenNet net_ = net; GenArgs a_ = a;
  static const bool nopack = getenv("ICRL_NO_XCD_PACK") != nullptr;
  int packed = nopack ? 0 : 1;
  void* params[] = {(void*)&net_, (void*)&a_, (void*)&pp, (void*)&packed};
  if (H > GENP_MAX_WGS) {      // 481..512-row batches: more workgroups than packed_grid leaves free on an XCD — only if two fit a CU
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gen_train_persistent_kernel, GENP_TH, lds) != hipSuccess || per_cu < 2) {
      (void)hipGetLastError();
      return -1;               // (the launch-per-phase form below the caller)
    }
  }
  e = hipLaunchCooperativeKernel((const void*)gen_train_persistent_kernel, dim3(packed ? XCD_STRIDE * (H - 1) + 1 : H), dim3(GENP_TH), params, (unsigned)lds, s);
  if (e == hipErrorCooperativeLaunchTooLarge && packed) {      // (a partition that cannot hold the sparse grid: the dense one, agent-scope stores)
    (void)hipGetLastError();
    packed = 0;
    e = hipLaunchCooperativeKernel((const void*)gen_train_persistent_kernel, dim3(H), dim3(GENP_TH), params, (unsigned)lds, s);
  }
  if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorNotSupported || e == hipErrorInvalidConfiguration) {
    (void)hipGetLastError();   // no co-resident grid on this device / partition: the three launches per optimiser step the header documents
    return -1;
  }
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gen_transpose_kernel, dim3((n + 255) / 256), dim3(256), 0, s, net, a.params, a.params_t);      // the forward / sampling kernels read params_t
  return (int)hipGetLastError();
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
