// The layer table of the generic-shape path (generic.hip) and the pieces other files need of it: the sampler of rollout.hip runs
// the same forward inside its persistent episode loop.  See generic.hip for the path itself.
#pragma once
#include "common.h"

namespace icrl {

constexpr int GEN_MAX_H = 256;                               // widest layer
#ifndef GEN_AHEAD
#define GEN_AHEAD 16                                         // operands fetched ahead of their dependent fmas (measured: 8 -> 27.0, 16 -> 22.4, 32 -> 23.9 us per forward+backward launch)
#endif
constexpr int GEN_MAX_DEPTH = 4;                             // layers of the shared trunk / of one branch
constexpr int GEN_MAX_LAYERS = 4 * GEN_MAX_DEPTH + 3;        // trunk + three branches + three heads
constexpr int GEN_MAX_STAGES = 2 * GEN_MAX_DEPTH + 1;
constexpr int GEN_MAX_ROW = 4 * GEN_MAX_DEPTH * GEN_MAX_H + MAX_ACT + 2;      // outputs of every layer of one row

// One Linear (+ tanh) of the network.  Layers are numbered in execution order; a layer reads the observation (in_buf = -1) or the
// output of an earlier layer, and writes act_off .. act_off + out_dim of the row's activation record.  The layers of one stage
// are independent and run side by side, one per slot (slot = threadIdx.x / W; the trunk on slot 0, branch r on slot r).
struct GenLayer { int in_dim, out_dim, w_off, b_off, in_buf, act_off, slot, tanh; };

struct GenNet {
  int O, A, discrete, log_std, n;          // log_std: parameter offset (-1 when discrete); n: parameter count
  int n_layers, n_stages, row_floats, W;   // W: threads per slot = the widest layer rounded up to 64
  int n_shared;                            // layers 0 .. n_shared-1 form the shared trunk (one per stage)
  int stage_begin[GEN_MAX_STAGES + 1];
  int head[3];                             // layer indices of action_net / value_net / cost_value_net (the last stage)
  GenLayer layer[GEN_MAX_LAYERS];
};

// icrl_policy_t -> GenNet.  Runs on the host (arch is host memory).  Returns 0 or the fail() code.
static inline int make_gen_net(const icrl_policy_t* p, GenNet* out, const char* who) {
  GenNet& g = *out;
  const int O = p->obs_dim, A = p->act_dim;
  if (O < 1 || O > 1024 || A < 1 || A > MAX_ACT) return fail("%s (generic path): obs_dim %d (1..1024), act_dim %d (1..%d)", who, O, A, MAX_ACT);
  int n_sh = 0, sh[GEN_MAX_DEPTH], n_br[3], br[3][GEN_MAX_DEPTH];
  if (p->arch == nullptr) {      // classic layout: three two-layer branches, widths padded to h1 = h2
    if (p->h1 != p->h2 || p->h1 % 64 != 0 || p->h1 < 64 || p->h1 > GEN_MAX_H)
      return fail("%s (generic path): padded hidden width %d x %d (equal, a multiple of 64, <= %d)", who, p->h1, p->h2, GEN_MAX_H);
    for (int r = 0; r < 3; ++r) { n_br[r] = 2; br[r][0] = p->h1; br[r][1] = p->h2; }
  } else {
    const int32_t* a = p->arch;
    n_sh = *a++;
    if (n_sh < 0 || n_sh > GEN_MAX_DEPTH) return fail("%s: %d shared layers (0..%d)", who, n_sh, GEN_MAX_DEPTH);
    for (int i = 0; i < n_sh; ++i) sh[i] = *a++;
    for (int r = 0; r < 3; ++r) {
      n_br[r] = *a++;
      if (n_br[r] < 0 || n_br[r] > GEN_MAX_DEPTH) return fail("%s: %d layers in branch %d (0..%d)", who, n_br[r], r, GEN_MAX_DEPTH);
      for (int i = 0; i < n_br[r]; ++i) br[r][i] = *a++;
    }
    for (int i = 0; i < n_sh; ++i) if (sh[i] < 1 || sh[i] > GEN_MAX_H) return fail("%s: shared layer %d has %d units (1..%d)", who, i, sh[i], GEN_MAX_H);
    for (int r = 0; r < 3; ++r)
      for (int i = 0; i < n_br[r]; ++i) if (br[r][i] < 1 || br[r][i] > GEN_MAX_H) return fail("%s: layer %d of branch %d has %d units (1..%d)", who, i, r, br[r][i], GEN_MAX_H);
  }
  g.O = O; g.A = A; g.discrete = p->discrete != 0;
  int off = 0, nl = 0, ns = 0, act = 0, widest = 64;
  if (g.discrete) g.log_std = -1; else { g.log_std = 0; off += A; }
  auto add = [&](int in_dim, int out_dim, int in_buf, int slot, int tanh, int w_off) {
    GenLayer& l = g.layer[nl];
    l.in_dim = in_dim; l.out_dim = out_dim; l.w_off = w_off; l.b_off = w_off + in_dim * out_dim; l.in_buf = in_buf; l.act_off = act; l.slot = slot; l.tanh = tanh;
    act += out_dim;
    if (out_dim > widest) widest = out_dim;
    return nl++;
  };
  // parameter order = the reference's state_dict: trunk, policy_net, value_net, cost_value_net (every layer W then b), then the heads
  int last = -1, last_dim = O;
  for (int i = 0; i < n_sh; ++i) {
    g.stage_begin[ns++] = nl;
    last = add(last_dim, sh[i], last, 0, 1, off);
    off += last_dim * sh[i] + sh[i];
    last_dim = sh[i];
  }
  int w_off[3][GEN_MAX_DEPTH], max_depth = 0;
  for (int r = 0; r < 3; ++r) {
    int d_in = last_dim;
    for (int i = 0; i < n_br[r]; ++i) { w_off[r][i] = off; off += d_in * br[r][i] + br[r][i]; d_in = br[r][i]; }
    if (n_br[r] > max_depth) max_depth = n_br[r];
  }
  int tip[3] = {last, last, last}, tip_dim[3] = {last_dim, last_dim, last_dim};
  for (int d = 0; d < max_depth; ++d) {
    g.stage_begin[ns++] = nl;
    for (int r = 0; r < 3; ++r)
      if (d < n_br[r]) { tip[r] = add(tip_dim[r], br[r][d], tip[r], r, 1, w_off[r][d]); tip_dim[r] = br[r][d]; }
  }
  g.stage_begin[ns++] = nl;
  for (int r = 0; r < 3; ++r) {
    const int n_out = r == 0 ? A : 1;
    g.head[r] = add(tip_dim[r], n_out, tip[r], r, 0, off);
    off += tip_dim[r] * n_out + n_out;
  }
  g.stage_begin[ns] = nl;
  g.n = off; g.n_layers = nl; g.n_stages = ns; g.row_floats = act; g.W = (widest + 63) / 64 * 64; g.n_shared = n_sh;
  if (p->n_params != g.n) return fail("%s: n_params = %d, the architecture needs %d", who, p->n_params, g.n);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// forward of ONE row: thread (slot, j) = unit j of the layer its slot runs in the current stage.  blockDim = 3 * W.
// ---------------------------------------------------------------------------------------------------------------
struct GenFwdShared {
  float x[1024];
  float act[GEN_MAX_ROW];
};

__device__ __forceinline__ int gen_my_layer(const GenNet& net, int stage, int slot) {
  for (int l = net.stage_begin[stage]; l < net.stage_begin[stage + 1]; ++l)
    if (net.layer[l].slot == slot) return l;
  return -1;
}

// PT: the per-layer transposes (gen_transpose_kernel): unit j reads Wt[k][j], consecutive units consecutive addresses — the row-major
// W[j][k] costs one cache line per lane and load.  Same fmaf chain (k ascending from the bias) either way.
__device__ __forceinline__ void gen_mlp_forward(const GenNet& net, const float* __restrict__ PT, const float* x, float* act, int slot, int j) {
  for (int s = 0; s < net.n_stages; ++s) {
    const int l = gen_my_layer(net, s, slot);
    if (l >= 0 && j < net.layer[l].out_dim) {
      const GenLayer& y = net.layer[l];
      const float* wt = PT + y.w_off + j;
      const float* in = y.in_buf < 0 ? x : act + net.layer[y.in_buf].act_off;
      const int n_in = y.in_dim, n_out = y.out_dim;
      float z = PT[y.b_off + j];
      int k = 0;
      for (; k + GEN_AHEAD <= n_in; k += GEN_AHEAD) {
        float w[GEN_AHEAD];
#pragma unroll
        for (int u = 0; u < GEN_AHEAD; ++u) w[u] = wt[(size_t)(k + u) * n_out];
#pragma unroll
        for (int u = 0; u < GEN_AHEAD; ++u) z = fmaf(w[u], in[k + u], z);
      }
      for (; k < n_in; ++k) z = fmaf(wt[(size_t)k * n_out], in[k], z);
      act[y.act_off + j] = y.tanh ? fast_tanh(z) : z;
    }
    __syncthreads();
  }
}

// The same forward with FOUR units per lane: wave `slot` runs its slot's layer of the stage, lane l the units 4 l .. 4 l + 3 — one 16-byte
// load per input from the transposed image (PT[k][4 l ..]: consecutive lanes, consecutive addresses), GEN_QAHEAD inputs in flight, four
// independent fmaf chains per lane.  Every unit's chain is gen_mlp_forward's (the bias, then its inputs in ascending order): bit-identical
// results, a quarter of the load instructions and of the threads (a 128-wide layer: 32 lanes).  The inputs are walked in FULLY UNROLLED
// blocks of 128 / 64 / 32 / 16 (then one by one): the compiler's wait-count bookkeeping gives up at a loop's back edge — a rolled loop of
// 16-input turns waited for ALL prefetched loads at the top of every turn (measured: ~125 cycles per input, the forward of the persistent
// rollout at 128-wide layers took 49 k cycles) —, inside a block every wait is vmcnt(GEN_QAHEAD - 1).  Workgroups of >= 192 threads; all call.
#ifndef GEN_QAHEAD
#define GEN_QAHEAD 16
#endif
typedef float gen_f4 __attribute__((ext_vector_type(4)));
template <int NB>
__device__ __forceinline__ void gen_quads_block(__amdgpu_buffer_rsrc_t rs, unsigned w0, unsigned stride, const float* in, int k0, gen_f4& z) {
  auto ld4 = [&](unsigned off) -> gen_f4 { return __builtin_bit_cast(gen_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0)); };
  constexpr int AH = NB < GEN_QAHEAD ? NB : GEN_QAHEAD;
  gen_f4 wv[AH];
#pragma unroll
  for (int u = 0; u < AH; ++u) wv[u] = ld4(w0 + stride * (unsigned)(k0 + u));
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const float xk = in[k0 + k];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = fmaf(wv[k % AH][i], xk, z[i]);
    if (k + AH < NB) wv[k % AH] = ld4(w0 + stride * (unsigned)(k0 + k + AH));
  }
}
__device__ __forceinline__ void gen_mlp_forward_quads(const GenNet& net, const float* __restrict__ PT, const float* x, float* act, int slot, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PT), 0, net.n * 4, 0x00020000);
  for (int s = 0; s < net.n_stages; ++s) {
    const int l = slot < 3 ? gen_my_layer(net, s, slot) : -1;
    if (l >= 0 && 4 * lane < net.layer[l].out_dim) {
      const GenLayer& y = net.layer[l];
      const float* in = y.in_buf < 0 ? x : act + net.layer[y.in_buf].act_off;
      const int n_in = y.in_dim, n_out = y.out_dim;
      const unsigned w0 = 4u * (unsigned)(y.w_off + 4 * lane), stride = 4u * (unsigned)n_out;      // bytes: PT[k * n_out + 4 lane ..]
      gen_f4 z = __builtin_bit_cast(gen_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(4u * (unsigned)(y.b_off + 4 * lane)), 0, 0));
      int k = 0;
      for (; k + 128 <= n_in; k += 128) gen_quads_block<128>(rs, w0, stride, in, k, z);
      if (k + 64 <= n_in) { gen_quads_block<64>(rs, w0, stride, in, k, z); k += 64; }
      if (k + 32 <= n_in) { gen_quads_block<32>(rs, w0, stride, in, k, z); k += 32; }
      if (k + 16 <= n_in) { gen_quads_block<16>(rs, w0, stride, in, k, z); k += 16; }
      if (k + 8 <= n_in) { gen_quads_block<8>(rs, w0, stride, in, k, z); k += 8; }
      if (k + 4 <= n_in) { gen_quads_block<4>(rs, w0, stride, in, k, z); k += 4; }
      if (k + 2 <= n_in) { gen_quads_block<2>(rs, w0, stride, in, k, z); k += 2; }
      if (k < n_in) gen_quads_block<1>(rs, w0, stride, in, k, z);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * lane + i < n_out) act[y.act_off + 4 * lane + i] = y.tanh ? fast_tanh(z[i]) : z[i];
    }
    __syncthreads();
  }
}

// The action distribution on the head outputs `out` of ONE row (policies.py:716-731 forward: sample / mode, clip, log-prob; :752-767
// evaluate_actions: `given` action, entropy).  The *_row pointers address this row (global memory or LDS), NULL = not wanted.
__device__ __forceinline__ void gen_policy_head(const GenNet& net, const float* __restrict__ P, const float* out, const float* noise_row,
                                                int deterministic, const float* alow, const float* ahigh, const float* given_row,
                                                float* actions_row, float* clipped_row, float& lp_out, float& ent_out) {
  const int A = net.A;
  float lp = 0.f, ent = 0.f;
  if (!net.discrete) {
    for (int o = 0; o < A; ++o) {
      const float ls = P[net.log_std + o], sd = __expf(ls), mean = out[o];
      float act = mean;
      if (given_row != nullptr) act = given_row[o];
      else if (!deterministic && noise_row != nullptr) act = mean + noise_row[o] * sd;      // Normal.rsample: loc + eps * scale
      const float diff = act - mean;
      lp += -(diff * diff) / (2.f * sd * sd) - ls - LOG_SQRT_2PI_F;
      ent += HALF_LOG_2PI_PLUS_HALF_F + ls;
      if (actions_row) actions_row[o] = act;
      if (clipped_row) clipped_row[o] = (alow != nullptr && ahigh != nullptr) ? fminf(fmaxf(act, alow[o]), ahigh[o]) : act;
    }
  } else {      // Categorical(logits): log-softmax, inverse-CDF sample on the injected uniform (spec: oracle/nets.py forward)
    float m = -INFINITY;
    for (int o = 0; o < A; ++o) m = fmaxf(m, out[o]);
    float se = 0.f;
    for (int o = 0; o < A; ++o) se += expf(out[o] - m);
    const float lse = m + logf(se);
    int action = 0;
    if (given_row != nullptr) action = (int)given_row[0];
    else if (deterministic || noise_row == nullptr) {
      float best = -1.f;
      for (int o = 0; o < A; ++o) { const float p = expf(out[o] - lse); if (p > best) { best = p; action = o; } }
    } else {
      const float u = noise_row[0];
      float cdf = 0.f;
      int cnt = 0;
      for (int o = 0; o < A; ++o) { cdf += expf(out[o] - lse); cnt += (u >= cdf) ? 1 : 0; }
      action = cnt < A - 1 ? cnt : A - 1;
    }
    lp = out[action] - lse;
    for (int o = 0; o < A; ++o) { const float lg = out[o] - lse; ent -= lg * expf(lg); }
    if (actions_row) actions_row[0] = (float)action;
    if (clipped_row) clipped_row[0] = (float)action;
  }
  lp_out = lp; ent_out = ent;
}

}  // namespace icrl
