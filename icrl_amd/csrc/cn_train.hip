// ConstraintNet backward step (zeta_theta training) — gfx950.
//
// ref: icrl/constraint_net.py:137-229 (train), :231-256 (compute_is_weights), :258-299 (prepare_data).
//
// Full-batch mode (batch_size None — the reference's default and every README configuration): one optimiser step per
// iteration over ALL nominal + expert rows (minibatch mode: see icrl_cn_train_minibatch below).  Per iteration:
//   cn_forward_kernel   blocks of 64 rows x 256 threads: ReLU-MLP + sigmoid forward, predictions and per-block partial sums
//   cn_finalize_kernel  one workgroup: fixed-order reduction of the partials, per-episode float32 products of the
//                       likelihood ratios (one wave per episode), KLs, importance weights, early-stop flag, losses
//   cn_backward_kernel  blocks of 64 rows: recompute the forward from LDS, analytic backward, per-block gradient partials
//   cn_adam_kernel      fixed-order sum of the partials over blocks + Adam (eps 1e-5, no gradient clipping)
// All reductions have a fixed order (no atomics), so repeated runs are bit-identical.
//
// Quirks kept: per-step weights act through mean(w) * mean(log zeta_N) (the reference's [B,1,1] x [B,1]
// broadcast); per-episode products are float32 and may overflow to inf / nan, in which case the comparisons of the
// early-stop test are false.
#include "common.h"

namespace icrl {

constexpr int CN_ROWS = 64;
constexpr int CN_TH = 256;
constexpr int CN_NPART = 16;   // statistics per block

constexpr int CN_MAX_LAYERS = 4;   // hidden layers (icrl_costnet_t: h1 .. h4)

struct CnDims {
  int D, nh, H[CN_MAX_LAYERS], HL;     // input, hidden layers, their widths, width of the last hidden layer
  int W[CN_MAX_LAYERS], b[CN_MAX_LAYERS], Wo, bo, n_params;      // state_dict order: W0 b0 W1 b1 ... Wo bo
  int wglobal;                 // 1: the weights stay in device memory (wide nets: the 64-row activation images alone fill the LDS)
  // LDS offsets (floats): weights, input image, one activation image per layer, TWO gradient images used alternately (layer l writes
  // sD[l & 1]; the parameter gradients of a layer are formed before its image is reused), logits / logit gradients
  int sW, sX, sH[CN_MAX_LAYERS], sD[2], sZ, total;
};

constexpr int CN_LDS_FLOATS = 160 * 1024 / 4;

__host__ __device__ inline CnDims make_cn_dims(int D, int nh, const int* H) {
  CnDims d;
  d.D = D; d.nh = nh;
  int off = 0, last = D, hmax = 0, hsum = 0;
  for (int l = 0; l < CN_MAX_LAYERS; ++l) {
    d.H[l] = l < nh ? H[l] : 0;
    d.W[l] = off; d.b[l] = off;
    if (l < nh) { d.b[l] = off + H[l] * last; off = d.b[l] + H[l]; last = H[l]; hsum += H[l] + 1; if (H[l] > hmax) hmax = H[l]; }
  }
  d.HL = last;
  d.Wo = off; off += last;
  d.bo = off; off += 1;
  d.n_params = off;
  const int act = CN_ROWS * (D + 1) + CN_ROWS * hsum + 2 * CN_ROWS * (hmax + 1) + 2 * CN_ROWS;
  d.wglobal = (d.n_params + act > CN_LDS_FLOATS) ? 1 : 0;      // the reference's widths (<= 64) keep their weights in LDS
  off = 0;
  d.sW = off; off += d.wglobal ? 0 : d.n_params;
  d.sX = off; off += CN_ROWS * (D + 1);
  for (int l = 0; l < CN_MAX_LAYERS; ++l) { d.sH[l] = off; off += l < nh ? CN_ROWS * (H[l] + 1) : 0; }
  d.sD[0] = off; off += CN_ROWS * (hmax + 1);
  d.sD[1] = off; off += CN_ROWS * (hmax + 1);
  d.sZ = off; off += 2 * CN_ROWS;
  d.total = off;
  return d;
}

// hidden widths of a descriptor as an array; 0 when the layer count is outside 0..CN_MAX_LAYERS (0: the sigmoid of one Linear)
__host__ __device__ inline int cn_widths(const icrl_costnet_t& cn, int* H) {
  H[0] = cn.h1; H[1] = cn.h2; H[2] = cn.h3; H[3] = cn.h4;
  return cn.n_hidden >= 0 && cn.n_hidden <= CN_MAX_LAYERS;
}

// scalars shared between the kernels of one train() call (device floats in `work`)
enum { SC_STOPPED = 0, SC_MEAN_W, SC_ITER, SC_MEAN_RATIO, SC_COUNT = 16 };

struct CnTrainArgs {
  CnDims d;
  float* params;
  float* exp_avg;
  float* exp_avg_sq;
  int* adam_t;
  const float* nominal;
  const float* expert;
  int Nn, Ne, n_ep, nb_n, nb_e;
  const int* ep_off;
  const int* row_ep;
  icrl_cn_hyper_t hp;
  float* start_preds;  // [Nn]
  float* preds_n;      // [Nn]
  float* preds_e;      // [Ne]
  float* part;         // [nb_n + nb_e][CN_NPART]
  float* ep_prod;      // [n_ep]
  float* ep_slog;      // [n_ep]
  float* normed;       // [n_ep]
  float* scal;         // [SC_COUNT]
  float* gpart;        // [nb_n + nb_e][n_params]
  float* metrics;      // [iterations][ICRL_CN_METRICS]
  const int* mb_idx;   // minibatch mode: the batch's row indices (the SAME rows of the nominal and the expert set) ...
  int mb_n;            // ... and their number; nb_n == nb_e == ceil(mb_n / 64) in the arguments of those launches
};

__device__ __forceinline__ void cn_block_mlp(const CnDims& d, float* sm, const float* params);

// block-level forward of CN_ROWS rows starting at row0 of `src` ([n, D]); leaves x, h1, (h2) in LDS, zeta in sm[sZ + row]
__device__ __forceinline__ void cn_block_forward(const CnDims& d, float* sm, const float* params, const float* src, int row0,
                                                 int n_rows_total, const int* gather = nullptr) {
  const int tid = threadIdx.x;
  const int D = d.D;
  for (int i = tid; i < CN_ROWS * D; i += CN_TH) {
    const int rr = i / D, k = i % D;
    float v = 0.f;
    if (row0 + rr < n_rows_total) v = src[(size_t)(gather ? gather[row0 + rr] : row0 + rr) * D + k];
    sm[d.sX + rr * (D + 1) + k] = v;
  }
  cn_block_mlp(d, sm, params);
}

// the ReLU-MLP + sigmoid of the CN_ROWS input rows already in sm[sX] (a barrier is taken first); leaves h1, (h2) in LDS, zeta in sm[sZ + row]
__device__ __forceinline__ void cn_block_mlp(const CnDims& d, float* sm, const float* params) {
  const int tid = threadIdx.x;
  const int row = tid & 63, part = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = d.D;
  if (!d.wglobal)
    for (int i = tid; i < d.n_params; i += CN_TH) sm[d.sW + i] = params[i];
  __syncthreads();
  const float* W = d.wglobal ? params : sm + d.sW;
  const float* in = sm + d.sX + row * (D + 1);
  int n_in = D;
  for (int l = 0; l < d.nh; ++l) {
    float* h = sm + d.sH[l] + row * (d.H[l] + 1);
    for (int j = part; j < d.H[l]; j += 4) {
      float acc = 0.f;
      const float* wr = W + d.W[l] + j * n_in;
      for (int k = 0; k < n_in; ++k) acc = fmaf(in[k], wr[k], acc);
      h[j] = fmaxf(acc + W[d.b[l] + j], 0.f);
    }
    __syncthreads();
    in = h; n_in = d.H[l];
  }
  if (part == 0) {
    float acc = 0.f;
    for (int j = 0; j < d.HL; ++j) acc = fmaf(in[j], W[d.Wo + j], acc);
    const float z = acc + W[d.bo];
    sm[d.sZ + row] = 1.f / (1.f + expf(-z));
  }
  __syncthreads();
}

__device__ __forceinline__ void cn_forward_body(const CnTrainArgs& a, int itr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (a.scal[SC_STOPPED] != 0.f) return;
  const CnDims& d = a.d;
  const int blk = blockIdx.x;
  const bool nominal = blk < a.nb_n;
  const int row0 = (nominal ? blk : blk - a.nb_n) * CN_ROWS;
  const int n_tot = nominal ? a.Nn : a.Ne;
  cn_block_forward(d, sm, a.params, nominal ? a.nominal : a.expert, row0, n_tot);
  if (threadIdx.x < 64) {
    const int row = threadIdx.x, g = row0 + row;
    const bool valid = g < n_tot;
    const float z = sm[d.sZ + row];
    const float eps = a.hp.eps;
    float ratio = 0.f;
    if (valid) {
      if (nominal) {
        a.preds_n[g] = z;
        if (itr == 0) a.start_preds[g] = z;
        const float zo = itr == 0 ? z : a.start_preds[g];
        ratio = (z + eps) / (zo + eps);
      } else {
        a.preds_e[g] = z;
      }
    }
    const float lg = valid ? logf(z + eps) : 0.f;
    // torch BCELoss clamps log at -100
    const float bce = valid ? (nominal ? fmaxf(logf(1.f - z), -100.f) : fmaxf(logf(z), -100.f)) : 0.f;
    const float s_log = wave_sum(lg);
    const float s_om = wave_sum(valid ? 1.f - z : 0.f);
    const float s_z = wave_sum(valid ? z : 0.f);
    const float mx = wave_max(valid ? z : -INFINITY);
    const float mn = wave_min(valid ? z : INFINITY);
    const float s_r = wave_sum(ratio);
    const float mxr = wave_max(valid ? ratio : -INFINITY);
    const float mnr = wave_min(valid ? ratio : INFINITY);
    const float s_b = wave_sum(bce);
    if (row == 0) {
      float* p = a.part + (size_t)blk * CN_NPART;
      p[0] = s_log; p[1] = s_om; p[2] = s_z; p[3] = mx; p[4] = mn; p[5] = s_r; p[6] = mxr; p[7] = mnr; p[8] = s_b;
    }
  }
}

__device__ __forceinline__ float wave_prod(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v *= __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ void cn_finalize_body(const CnTrainArgs& a, int itr) {
  __shared__ float red[2][CN_NPART];
  if (a.scal[SC_STOPPED] != 0.f) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float eps = a.hp.eps;
  // ---- fixed-order reduction of the per-block partials: wave <-> (set, stat) pair, lanes stride over the blocks
  for (int pair = wv; pair < 2 * 9; pair += 16) {
    const int set = pair / 9, s = pair % 9;
    const int b0 = set == 0 ? 0 : a.nb_n, b1 = set == 0 ? a.nb_n : a.nb_n + a.nb_e;
    const bool is_max = (s == 3 || s == 6), is_min = (s == 4 || s == 7);
    float acc = is_max ? -INFINITY : (is_min ? INFINITY : 0.f);
    for (int b = b0 + lane; b < b1; b += 64) {
      const float v = a.part[(size_t)b * CN_NPART + s];
      if (is_max) acc = fmaxf(acc, v);
      else if (is_min) acc = fminf(acc, v);
      else acc += v;
    }
    acc = is_max ? wave_max(acc) : (is_min ? wave_min(acc) : wave_sum(acc));
    if (lane == 0) red[set][s] = acc;
  }
  // ---- per-episode float32 product of the ratios and sum of log(zeta + eps): one wave per episode
  if (a.hp.importance_sampling) {
    for (int e = wv; e < a.n_ep; e += 16) {
      float pr = 1.f, sl = 0.f;
      for (int i = a.ep_off[e] + lane; i < a.ep_off[e + 1]; i += 64) {
        const float z = a.preds_n[i];
        const float zo = itr == 0 ? z : a.start_preds[i];
        pr *= (z + eps) / (zo + eps);
        sl += logf(z + eps);
      }
      pr = wave_prod(pr);
      sl = wave_sum(sl);
      if (lane == 0) { a.ep_prod[e] = pr; a.ep_slog[e] = sl; }
    }
  }
  __syncthreads();
  if (tid != 0) return;
  const float Nn = (float)a.Nn, Ne = (float)a.Ne;
  float* m = a.metrics + (size_t)itr * ICRL_CN_METRICS;
  float kl_on = 0.f, kl_no = 0.f, is_mean = 1.f, is_max = 1.f, is_min = 1.f;
  float weighted = red[0][0] / Nn;   // nominal_loss for the no-IS case: mean(log(zeta_N + eps))
  bool stop = false;
  if (a.hp.importance_sampling) {
    const int ne = a.n_ep;
    float sum_p = 0.f;
    for (int e = 0; e < ne; ++e) sum_p += a.ep_prod[e];
    const float pm = sum_p / (float)ne;
    float s1 = 0.f, s2 = 0.f;
    for (int e = 0; e < ne; ++e) {
      const float lp = logf(a.ep_prod[e] + eps);
      s1 += -lp;
      s2 += (a.ep_prod[e] - pm) * lp / (pm + eps);
    }
    kl_on = s1 / (float)ne;
    kl_no = s2 / (float)ne;
    if (a.hp.per_step) {
      const float mean_ratio = red[0][5] / Nn;
      a.scal[SC_MEAN_RATIO] = mean_ratio;
      is_mean = (red[0][5] / mean_ratio) / Nn;
      is_max = red[0][6] / mean_ratio;
      is_min = red[0][7] / mean_ratio;
      weighted = is_mean * (red[0][0] / Nn);        // mean over the [B,B,1] broadcast == mean(w) * mean(log)
      a.scal[SC_MEAN_W] = is_mean;
    } else {
      float acc = 0.f, wsum = 0.f, wmax = -INFINITY, wmin = INFINITY;
      for (int e = 0; e < ne; ++e) {
        const float nw = (float)ne * a.ep_prod[e] / (sum_p + eps);
        a.normed[e] = nw;
        acc += nw * a.ep_slog[e];
        wsum += nw * (float)(a.ep_off[e + 1] - a.ep_off[e]);
        wmax = fmaxf(wmax, nw); wmin = fminf(wmin, nw);
      }
      weighted = acc / Nn;
      is_mean = wsum / Nn; is_max = wmax; is_min = wmin;
    }
    stop = (a.hp.target_kl_old_new != -1.f && kl_on > a.hp.target_kl_old_new) ||
           (a.hp.target_kl_new_old != -1.f && kl_no > a.hp.target_kl_new_old);
  }
  const float expert_loss = a.hp.gail ? -(red[1][8] / Ne) : red[1][0] / Ne;
  const float unweighted = red[0][0] / Nn;
  float nominal_loss, reg, loss;
  if (a.hp.gail) {
    nominal_loss = -(red[0][8] / Nn);
    reg = 0.f;
    loss = nominal_loss + expert_loss;
  } else {
    nominal_loss = weighted;
    reg = a.hp.reg_coeff * (red[1][1] / Ne + red[0][1] / Nn);
    loss = (-expert_loss + nominal_loss) + reg;
  }
  m[0] = stop ? 1.f : 0.f; m[1] = kl_on; m[2] = kl_no; m[3] = is_mean; m[4] = is_max; m[5] = is_min;
  m[6] = loss; m[7] = expert_loss; m[8] = unweighted; m[9] = nominal_loss; m[10] = reg;
  m[11] = red[0][3]; m[12] = red[0][4]; m[13] = red[0][2] / Nn;
  m[14] = red[1][3]; m[15] = red[1][4]; m[16] = red[1][2] / Ne;
  m[17] = 0.f;
  if (stop) a.scal[SC_STOPPED] = 1.f;
}


// =================================================================================================================
// minibatch mode (cn_batch_size; ref: constraint_net.py:181-206,300-316): after the importance weights / early-stop test of
// the iteration (cn_forward_kernel + cn_finalize_kernel over ALL rows, as above), one optimiser step per batch of the
// permutation; the batch takes the SAME row indices from the nominal and the expert set.
// =================================================================================================================
// importance weight of nominal row i as fixed at the start of the iteration
__device__ __forceinline__ float cn_row_weight(const CnTrainArgs& a, int i) {
  if (!a.hp.importance_sampling) return 1.f;
  if (a.hp.per_step) return ((a.preds_n[i] + a.hp.eps) / (a.start_preds[i] + a.hp.eps)) / a.scal[SC_MEAN_RATIO];
  return a.normed[a.row_ep[i]];
}

__global__ void __launch_bounds__(CN_TH) cn_mb_forward_kernel(CnTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (a.scal[SC_STOPPED] != 0.f) return;
  const CnDims& d = a.d;
  const int blk = blockIdx.x;
  const bool nominal = blk < a.nb_n;
  const int row0 = (nominal ? blk : blk - a.nb_n) * CN_ROWS;
  cn_block_forward(d, sm, a.params, nominal ? a.nominal : a.expert, row0, a.mb_n, a.mb_idx);
  if (threadIdx.x < 64) {
    const int row = threadIdx.x, g = row0 + row;
    const bool valid = g < a.mb_n;
    const float z = sm[d.sZ + row], eps = a.hp.eps;
    const float lg = valid ? logf(z + eps) : 0.f;
    const float w = (valid && nominal) ? cn_row_weight(a, a.mb_idx[g]) : 0.f;
    const float bce = valid ? (nominal ? fmaxf(logf(1.f - z), -100.f) : fmaxf(logf(z), -100.f)) : 0.f;
    const float s_log = wave_sum(lg);
    const float s_om = wave_sum(valid ? 1.f - z : 0.f);
    const float s_z = wave_sum(valid ? z : 0.f);
    const float mx = wave_max(valid ? z : -INFINITY);
    const float mn = wave_min(valid ? z : INFINITY);
    const float s_w = wave_sum(w);
    const float s_wl = wave_sum(w * lg);
    const float s_b = wave_sum(bce);
    if (row == 0) {
      float* p = a.part + (size_t)blk * CN_NPART;
      p[0] = s_log; p[1] = s_om; p[2] = s_z; p[3] = mx; p[4] = mn; p[5] = s_w; p[6] = s_wl; p[7] = 0.f; p[8] = s_b;
    }
  }
}

__global__ void __launch_bounds__(64) cn_mb_finalize_kernel(CnTrainArgs a, int itr) {
  if (a.scal[SC_STOPPED] != 0.f) return;
  const int lane = threadIdx.x;
  float red[2][9];
  for (int set = 0; set < 2; ++set)
    for (int s = 0; s < 9; ++s) {
      const int b0 = set == 0 ? 0 : a.nb_n, b1 = set == 0 ? a.nb_n : a.nb_n + a.nb_e;
      const bool is_max = s == 3, is_min = s == 4;
      float acc = is_max ? -INFINITY : (is_min ? INFINITY : 0.f);
      for (int b = b0 + lane; b < b1; b += 64) {
        const float v = a.part[(size_t)b * CN_NPART + s];
        if (is_max) acc = fmaxf(acc, v);
        else if (is_min) acc = fminf(acc, v);
        else acc += v;
      }
      red[set][s] = is_max ? wave_max(acc) : (is_min ? wave_min(acc) : wave_sum(acc));
    }
  if (lane != 0) return;
  const float n = (float)a.mb_n;
  float* m = a.metrics + (size_t)itr * ICRL_CN_METRICS;
  const float unweighted = red[0][0] / n;
  float nominal_loss, expert_loss, reg, loss;
  if (a.hp.gail) {
    nominal_loss = -(red[0][8] / n);
    expert_loss = -(red[1][8] / n);
    reg = 0.f;
    loss = nominal_loss + expert_loss;
  } else {
    expert_loss = red[1][0] / n;
    const float mean_w = red[0][5] / n;
    // per-step weights are [b,1,1] against [b,1] predictions: the mean over the broadcast is mean(w) * mean(log zeta)
    nominal_loss = (a.hp.importance_sampling && a.hp.per_step) ? mean_w * unweighted : red[0][6] / n;
    a.scal[SC_MEAN_W] = mean_w;
    reg = a.hp.reg_coeff * (red[1][1] / n + red[0][1] / n);
    loss = (-expert_loss + nominal_loss) + reg;
  }
  m[6] = loss; m[7] = expert_loss; m[8] = unweighted; m[9] = nominal_loss; m[10] = reg;
  m[11] = red[0][3]; m[12] = red[0][4]; m[13] = red[0][2] / n;
  m[14] = red[1][3]; m[15] = red[1][4]; m[16] = red[1][2] / n;
}

template <bool MB>
__device__ __forceinline__ void cn_backward_body(const CnTrainArgs& a, int itr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (a.scal[SC_STOPPED] != 0.f) return;
  const CnDims& d = a.d;
  const int blk = blockIdx.x, tid = threadIdx.x;
  const bool nominal = blk < a.nb_n;
  const int row0 = (nominal ? blk : blk - a.nb_n) * CN_ROWS;
  const int n_tot = MB ? a.mb_n : (nominal ? a.Nn : a.Ne);
  const float cnt_n = (float)(MB ? a.mb_n : a.Nn), cnt_e = (float)(MB ? a.mb_n : a.Ne);
  cn_block_forward(d, sm, a.params, nominal ? a.nominal : a.expert, row0, n_tot, MB ? a.mb_idx : nullptr);
  const int row = tid & 63, part = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* W = d.wglobal ? a.params : sm + d.sW;
  // ---- d loss / d logit
  if (part == 0) {
    const int g = row0 + row;
    float dz = 0.f;
    if (g < n_tot) {
      const float z = sm[d.sZ + row], eps = a.hp.eps;
      float dzeta;
      if (a.hp.gail) {
        // BCE(zeta_N, 0) = -mean(max(log(1 - zeta), -100)); BCE(zeta_E, 1) = -mean(max(log zeta, -100))
        if (nominal) dzeta = (logf(1.f - z) > -100.f) ? 1.f / (cnt_n * (1.f - z)) : 0.f;
        else dzeta = (logf(z) > -100.f) ? -1.f / (cnt_e * z) : 0.f;
      } else if (nominal) {
        float wgt = 1.f;
        if (a.hp.importance_sampling) wgt = a.hp.per_step ? a.scal[SC_MEAN_W] : a.normed[a.row_ep[MB ? a.mb_idx[g] : g]];
        dzeta = wgt / (cnt_n * (z + eps)) - a.hp.reg_coeff / cnt_n;
      } else {
        dzeta = -1.f / (cnt_e * (z + eps)) - a.hp.reg_coeff / cnt_e;
      }
      dz = dzeta * (z * (1.f - z));
    }
    sm[d.sZ + CN_ROWS + row] = dz;
  }
  __syncthreads();
  const float dz = sm[d.sZ + CN_ROWS + row];
  float* gp = a.gpart + (size_t)blk * d.n_params;
  // ---- per-block parameter gradients: one thread per parameter, rows added in order.  Output layer first (logit gradients x last
  // hidden image), then layer by layer from the top: the layer's pre-activation gradients, then the gradients of its parameters
  {
    const int hb = d.nh > 0 ? d.sH[d.nh - 1] : d.sX, hs = d.HL + 1;      // (no hidden layer: the input image, HL = D)
    for (int p = tid; p < d.HL + 1; p += CN_TH) {
      float acc = 0.f;
      if (p < d.HL) { for (int rr = 0; rr < CN_ROWS; ++rr) acc = fmaf(sm[d.sZ + CN_ROWS + rr], sm[hb + rr * hs + p], acc); }     // Wo[j]
      else { for (int rr = 0; rr < CN_ROWS; ++rr) acc += sm[d.sZ + CN_ROWS + rr]; }                                            // bo
      gp[d.Wo + p] = acc;
    }
  }
  for (int l = d.nh - 1; l >= 0; --l) {
    const int Hl = d.H[l];
    const float* h = sm + d.sH[l] + row * (Hl + 1);
    float* dl = sm + d.sD[l & 1] + row * (Hl + 1);
    if (l == d.nh - 1) {
      for (int j = part; j < Hl; j += 4) dl[j] = h[j] > 0.f ? dz * W[d.Wo + j] : 0.f;
    } else {
      const int Hu = d.H[l + 1];
      const float* du = sm + d.sD[(l + 1) & 1] + row * (Hu + 1);
      for (int k = part; k < Hl; k += 4) {
        float acc = 0.f;
        for (int j = 0; j < Hu; ++j) acc = fmaf(du[j], W[d.W[l + 1] + j * Hl + k], acc);
        dl[k] = h[k] > 0.f ? acc : 0.f;
      }
    }
    __syncthreads();
    const int n_in = l == 0 ? d.D : d.H[l - 1];
    const int ib = l == 0 ? d.sX : d.sH[l - 1], is = n_in + 1, db = d.sD[l & 1], ds = Hl + 1;
    for (int p = tid; p < Hl * n_in + Hl; p += CN_TH) {
      float acc = 0.f;
      if (p < Hl * n_in) {                              // W_l[j][k]
        const int j = p / n_in, k = p % n_in;
        for (int rr = 0; rr < CN_ROWS; ++rr) acc = fmaf(sm[db + rr * ds + j], sm[ib + rr * is + k], acc);
      } else {                                          // b_l[j]
        const int j = p - Hl * n_in;
        for (int rr = 0; rr < CN_ROWS; ++rr) acc += sm[db + rr * ds + j];
      }
      gp[d.W[l] + p] = acc;
    }
  }
}

__device__ __forceinline__ void cn_adam_body(const CnTrainArgs& a, int itr, int upd) {
  if (a.scal[SC_STOPPED] != 0.f) return;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = a.nb_n + a.nb_e;
  const int t = a.adam_t[0] + upd + 1;   // upd: optimiser steps before this one in the call; adam_t is advanced once at the end
  if (p < a.d.n_params) {
    float g = 0.f;
    for (int b = 0; b < nb; ++b) g += a.gpart[(size_t)b * a.d.n_params + p];
    const double b1 = a.hp.adam_beta1, b2 = a.hp.adam_beta2;
    const float w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2);
    float m = a.exp_avg[p], v = a.exp_avg_sq[p];
    m = m + (g - m) * w1;
    v = v * a.hp.adam_beta2 + w2 * (g * g);
    const float step_size = (float)((double)a.hp.lr / (1.0 - pow(b1, (double)t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)t));
    const float denom = sqrtf(v) / bc2_sqrt + a.hp.adam_eps;
    a.params[p] = a.params[p] - step_size * (m / denom);
    a.exp_avg[p] = m; a.exp_avg_sq[p] = v;
  }
  if (p == 0) { a.metrics[(size_t)itr * ICRL_CN_METRICS + 17] = 1.f; a.scal[SC_ITER] = (float)(upd + 1); }
}

// ---- the kernels proper: single-run forms (arguments by value) and batched forms (several independent constraint nets of one
// shape in one launch: run = blockIdx.y, argument blocks in device memory; grids are sized for the largest run, surplus workgroups
// of smaller runs leave at once)
__global__ void __launch_bounds__(CN_TH) cn_forward_kernel(CnTrainArgs a, int itr) { cn_forward_body(a, itr); }
__global__ void __launch_bounds__(1024) cn_finalize_kernel(CnTrainArgs a, int itr) { cn_finalize_body(a, itr); }
template <bool MB>
__global__ void __launch_bounds__(CN_TH) cn_backward_kernel(CnTrainArgs a, int itr) { cn_backward_body<MB>(a, itr); }
__global__ void __launch_bounds__(256) cn_adam_kernel(CnTrainArgs a, int itr, int upd) { cn_adam_body(a, itr, upd); }
__global__ void cn_commit_kernel(CnTrainArgs a) {
  // advance the optimiser step counter by the number of executed updates
  if (threadIdx.x == 0 && blockIdx.x == 0) a.adam_t[0] += (int)a.scal[SC_ITER];
}

__global__ void __launch_bounds__(CN_TH) cn_forward_batch_kernel(const CnTrainArgs* __restrict__ runs, int itr) {
  const CnTrainArgs a = runs[blockIdx.y];
  if ((int)blockIdx.x >= a.nb_n + a.nb_e || itr >= a.hp.iterations) return;
  cn_forward_body(a, itr);
}
__global__ void __launch_bounds__(1024) cn_finalize_batch_kernel(const CnTrainArgs* __restrict__ runs, int itr) {
  const CnTrainArgs a = runs[blockIdx.y];
  if (itr >= a.hp.iterations) return;
  cn_finalize_body(a, itr);
}
__global__ void __launch_bounds__(CN_TH) cn_backward_batch_kernel(const CnTrainArgs* __restrict__ runs, int itr) {
  const CnTrainArgs a = runs[blockIdx.y];
  if ((int)blockIdx.x >= a.nb_n + a.nb_e || itr >= a.hp.iterations) return;
  cn_backward_body<false>(a, itr);
}
__global__ void __launch_bounds__(256) cn_adam_batch_kernel(const CnTrainArgs* __restrict__ runs, int itr) {
  const CnTrainArgs a = runs[blockIdx.y];
  if (itr >= a.hp.iterations) return;
  cn_adam_body(a, itr, itr);
}
__global__ void cn_commit_batch_kernel(const CnTrainArgs* __restrict__ runs) {
  const CnTrainArgs a = runs[blockIdx.y];
  if (threadIdx.x == 0 && blockIdx.x == 0) a.adam_t[0] += (int)a.scal[SC_ITER];
}

__global__ void __launch_bounds__(64) cn_prepare_kernel(icrl_costnet_t cn, const double* obs, const float* acs, int N, float* out) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const double* obs_row = obs + (size_t)n * cn.obs_dim;
  const float* acs_row = acs + (size_t)n * (cn.is_discrete ? 1 : cn.acs_dim);
  for (int i = lane; i < cn.in_dim; i += 64) {
    const int sel = cn.select_dim[i];
    float v;
    if (sel < cn.obs_dim) {
      double o = obs_row[sel];
      if (cn.obs_mean != nullptr && cn.obs_var != nullptr) o = (o - cn.obs_mean[sel]) / sqrt(cn.obs_var[sel] + cn.eps);
      if (cn.clip_obs >= 0.0) o = fmin(fmax(o, -cn.clip_obs), cn.clip_obs);
      v = (float)o;
    } else {
      const int k = sel - cn.obs_dim;
      float x = cn.is_discrete ? (((int)acs_row[0] == k) ? 1.f : 0.f) : acs_row[k];
      if (cn.action_low != nullptr && cn.action_high != nullptr) x = fminf(fmaxf(x, cn.action_low[k]), cn.action_high[k]);
      v = x;
    }
    out[(size_t)n * cn.in_dim + i] = v;
  }
}

// ConstraintNet.cost_function / the GAIL discriminator's reward for nets the one-wave-per-row kernel of rollout.hip does not hold
// (a hidden layer above 64 units): 64 rows per workgroup, prepare_data (constraint_net.py:258-299) straight into the LDS input image,
// then the block MLP of the update kernels.  mode 0: cost = 1 - zeta; 1: zeta; 2: log(zeta + eps).
__global__ void __launch_bounds__(CN_TH) cn_cost_rows_kernel(icrl_costnet_t cn, CnDims d, const double* obs, const float* acs, int N, float* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, row0 = blockIdx.x * CN_ROWS, D = d.D;
  const int AS = cn.is_discrete ? 1 : cn.acs_dim;
  for (int i = tid; i < CN_ROWS * D; i += CN_TH) {
    const int rr = i / D, k = i % D, n = row0 + rr;
    float v = 0.f;
    if (n < N) {
      const int sel = cn.select_dim[k];
      if (sel < cn.obs_dim) {
        double o = obs[(size_t)n * cn.obs_dim + sel];
        if (cn.obs_mean != nullptr && cn.obs_var != nullptr) o = (o - cn.obs_mean[sel]) / sqrt(cn.obs_var[sel] + cn.eps);
        if (cn.clip_obs >= 0.0) o = fmin(fmax(o, -cn.clip_obs), cn.clip_obs);
        v = (float)o;
      } else {
        const int kk = sel - cn.obs_dim;
        float x = cn.is_discrete ? (((int)acs[(size_t)n * AS] == kk) ? 1.f : 0.f) : acs[(size_t)n * AS + kk];
        if (cn.action_low != nullptr && cn.action_high != nullptr) x = fminf(fmaxf(x, cn.action_low[kk]), cn.action_high[kk]);
        v = x;
      }
    }
    sm[d.sX + rr * (D + 1) + k] = v;
  }
  cn_block_mlp(d, sm, cn.params);
  if (tid < CN_ROWS && row0 + tid < N) {
    const float zeta = sm[d.sZ + tid];
    out[row0 + tid] = mode == 1 ? zeta : (mode == 2 ? logf(zeta + (float)cn.eps) : 1.f - zeta);
  }
}

// descriptor -> CnDims + dynamic LDS bytes of the 64-row kernels; refuses (fail()) what they do not hold
static int cn_dims_checked(const icrl_costnet_t* cn, const char* who, CnDims* d, size_t* lds) {
  int H[CN_MAX_LAYERS];
  if (!cn_widths(*cn, H) || cn->in_dim < 1) return fail("%s: %d hidden layers (0..%d), in_dim %d", who, cn->n_hidden, CN_MAX_LAYERS, cn->in_dim);
  for (int l = 0; l < cn->n_hidden; ++l)
    if (H[l] < 1) return fail("%s: hidden layer %d has %d units", who, l, H[l]);
  *d = make_cn_dims(cn->in_dim, cn->n_hidden, H);
  if (d->n_params != cn->n_params)
    return fail("%s: n_params = %d but in_dim %d / hidden (%d, %d, %d, %d) x %d layers need %d", who, cn->n_params, cn->in_dim, cn->h1, cn->h2, cn->h3, cn->h4,
                cn->n_hidden, d->n_params);
  *lds = (size_t)d->total * sizeof(float);
  if (*lds > 160 * 1024)
    return fail("%s: the 64-row activation images of in_dim %d / hidden (%d, %d, %d, %d) x %d layers take %zu B of LDS, 160 KB available (e.g. 2 x 128, 3 x 96, "
                "4 x 64 units fit)", who, cn->in_dim, cn->h1, cn->h2, cn->h3, cn->h4, cn->n_hidden, *lds);
  return 0;
}

int launch_cn_cost_rows(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* out, int mode, hipStream_t s) {
  CnDims d;
  size_t lds;
  if (int e = cn_dims_checked(cn, "cost forward", &d, &lds)) return e;
  hipError_t e = hipFuncSetAttribute((const void*)cn_cost_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(cn_cost_rows_kernel, dim3((N + CN_ROWS - 1) / CN_ROWS), dim3(CN_TH), lds, s, *cn, d, obs, acs, N, out, mode);
  return (int)hipGetLastError();
}

static void cn_work_layout(int n_params, int Nn, int Ne, int n_ep, size_t* offs /* 9 */, size_t* total) {
  const size_t nb = (size_t)((Nn + CN_ROWS - 1) / CN_ROWS + (Ne + CN_ROWS - 1) / CN_ROWS);
  size_t o = 0;
  offs[0] = o; o += (size_t)Nn;            // start_preds
  offs[1] = o; o += (size_t)Nn;            // preds_n
  offs[2] = o; o += (size_t)Ne;            // preds_e
  offs[3] = o; o += nb * CN_NPART;         // part
  offs[4] = o; o += (size_t)n_ep;          // ep_prod
  offs[5] = o; o += (size_t)n_ep;          // ep_slog
  offs[6] = o; o += (size_t)n_ep;          // normed
  offs[7] = o; o += SC_COUNT;              // scal
  offs[8] = o; o += nb * (size_t)n_params; // gpart
  *total = o;
}

}  // namespace icrl

using namespace icrl;

extern "C" size_t icrl_cn_train_work_floats(int n_params, int Nn, int Ne, int n_ep) {
  size_t offs[9], total;
  cn_work_layout(n_params, Nn, Ne, n_ep, offs, &total);
  return total;
}

extern "C" int icrl_cn_prepare(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* out, void* stream) {
  if (N <= 0) return fail("icrl_cn_prepare: N = %d rows", N);
  hipLaunchKernelGGL(cn_prepare_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, *cn, obs, acs, N, out);
  return (int)hipGetLastError();
}

// argument checks + argument block of one full-batch train() call; *lds = dynamic LDS bytes of the forward / backward kernels
static int make_cn_train_args(const icrl_costnet_t* cn, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, const float* nominal,
                              const float* expert, int Nn, int Ne, const int32_t* ep_offsets, const int32_t* row_episode, int n_ep,
                              const icrl_cn_hyper_t* hp, float* work, float* metrics, CnTrainArgs& a, size_t* lds_out) {
  if (Nn <= 0 || Ne <= 0 || n_ep <= 0 || hp->iterations < 0)
    return fail("icrl_cn_train: needs nominal rows (%d), expert rows (%d), episodes (%d) > 0 and iterations (%d) >= 0", Nn, Ne, n_ep, hp->iterations);
  size_t lds;
  if (int e = cn_dims_checked(cn, "icrl_cn_train", &a.d, &lds)) return e;
  *lds_out = lds;
  a.params = cn->params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.adam_t = adam_step;
  a.nominal = nominal; a.expert = expert; a.Nn = Nn; a.Ne = Ne; a.n_ep = n_ep;
  a.nb_n = (Nn + CN_ROWS - 1) / CN_ROWS; a.nb_e = (Ne + CN_ROWS - 1) / CN_ROWS;
  a.ep_off = ep_offsets; a.row_ep = row_episode; a.hp = *hp;
  size_t offs[9], total;
  cn_work_layout(cn->n_params, Nn, Ne, n_ep, offs, &total);
  a.start_preds = work + offs[0]; a.preds_n = work + offs[1]; a.preds_e = work + offs[2]; a.part = work + offs[3];
  a.ep_prod = work + offs[4]; a.ep_slog = work + offs[5]; a.normed = work + offs[6]; a.scal = work + offs[7];
  a.gpart = work + offs[8]; a.metrics = metrics; a.mb_idx = nullptr; a.mb_n = 0;
  return 0;
}

extern "C" int icrl_cn_train(const icrl_costnet_t* cn, float* exp_avg, float* exp_avg_sq, int32_t* adam_step,
                             const float* nominal, const float* expert, int Nn, int Ne, const int32_t* ep_offsets,
                             const int32_t* row_episode, int n_ep, const icrl_cn_hyper_t* hp, float* work, float* metrics,
                             void* stream) {
  CnTrainArgs a;
  size_t lds = 0;
  const int bad = make_cn_train_args(cn, exp_avg, exp_avg_sq, adam_step, nominal, expert, Nn, Ne, ep_offsets, row_episode, n_ep, hp, work,
                                     metrics, a, &lds);
  if (bad) return bad;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a.scal, 0, SC_COUNT * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  if (hp->iterations > 0) {
    e = hipMemsetAsync(metrics, 0, (size_t)hp->iterations * ICRL_CN_METRICS * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  e = hipFuncSetAttribute((const void*)cn_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute((const void*)cn_backward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int nb = a.nb_n + a.nb_e;
  for (int itr = 0; itr < hp->iterations; ++itr) {
    hipLaunchKernelGGL(cn_forward_kernel, dim3(nb), dim3(CN_TH), lds, s, a, itr);
    hipLaunchKernelGGL(cn_finalize_kernel, dim3(1), dim3(1024), 0, s, a, itr);
    hipLaunchKernelGGL(cn_backward_kernel<false>, dim3(nb), dim3(CN_TH), lds, s, a, itr);
    hipLaunchKernelGGL(cn_adam_kernel, dim3((cn->n_params + 255) / 256), dim3(256), 0, s, a, itr, itr);
  }
  hipLaunchKernelGGL(cn_commit_kernel, dim3(1), dim3(64), 0, s, a);
  return (int)hipGetLastError();
}

extern "C" int icrl_cn_train_batch(int n_runs, const icrl_cn_train_job_t* jobs, void* args_ws, long long args_ws_bytes, void* stream) {
  static_assert(sizeof(CnTrainArgs) <= ICRL_BATCH_ARGS_BYTES, "ICRL_BATCH_ARGS_BYTES");
  if (n_runs < 1 || n_runs > 65535) return fail("icrl_cn_train_batch: n_runs = %d (1..65535)", n_runs);
  if (args_ws == nullptr || args_ws_bytes < (long long)n_runs * ICRL_BATCH_ARGS_BYTES)
    return fail("icrl_cn_train_batch: args_ws holds %lld B, %d runs need %lld", args_ws_bytes, n_runs, (long long)n_runs * ICRL_BATCH_ARGS_BYTES);
  hipStream_t s = (hipStream_t)stream;
  CnTrainArgs* d_args = (CnTrainArgs*)args_ws;
  size_t lds0 = 0;
  int nb_max = 0, iters_max = 0, np0 = 0;
  for (int r = 0; r < n_runs; ++r) {
    const icrl_cn_train_job_t& j = jobs[r];
    CnTrainArgs a;
    size_t lds = 0;
    const int bad = make_cn_train_args(j.cn, j.exp_avg, j.exp_avg_sq, j.adam_step, j.nominal, j.expert, j.Nn, j.Ne, j.ep_offsets,
                                       j.row_episode, j.n_ep, j.hp, j.work, j.metrics, a, &lds);
    if (bad) return bad;
    if (r == 0) { lds0 = lds; np0 = a.d.n_params; }
    else if (lds != lds0 || a.d.n_params != np0)
      return fail("icrl_cn_train_batch: run %d's network shape differs from run 0's (the runs of a batch share one grid)", r);
    nb_max = a.nb_n + a.nb_e > nb_max ? a.nb_n + a.nb_e : nb_max;
    iters_max = j.hp->iterations > iters_max ? j.hp->iterations : iters_max;
    hipError_t e = hipMemsetAsync(a.scal, 0, SC_COUNT * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    if (j.hp->iterations > 0) {
      e = hipMemsetAsync(j.metrics, 0, (size_t)j.hp->iterations * ICRL_CN_METRICS * sizeof(float), s);
      if (e != hipSuccess) return (int)e;
    }
    const int pe = put_args(a, d_args + r, s);
    if (pe) return pe;
  }
  hipError_t e = hipFuncSetAttribute((const void*)cn_forward_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute((const void*)cn_backward_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
  if (e != hipSuccess) return (int)e;
  for (int itr = 0; itr < iters_max; ++itr) {
    hipLaunchKernelGGL(cn_forward_batch_kernel, dim3(nb_max, n_runs), dim3(CN_TH), lds0, s, d_args, itr);
    hipLaunchKernelGGL(cn_finalize_batch_kernel, dim3(1, n_runs), dim3(1024), 0, s, d_args, itr);
    hipLaunchKernelGGL(cn_backward_batch_kernel, dim3(nb_max, n_runs), dim3(CN_TH), lds0, s, d_args, itr);
    hipLaunchKernelGGL(cn_adam_batch_kernel, dim3((np0 + 255) / 256, n_runs), dim3(256), 0, s, d_args, itr);
  }
  hipLaunchKernelGGL(cn_commit_batch_kernel, dim3(1, n_runs), dim3(64), 0, s, d_args);
  return (int)hipGetLastError();
}

extern "C" int icrl_cn_train_minibatch(const icrl_costnet_t* cn, float* exp_avg, float* exp_avg_sq, int32_t* adam_step,
                                       const float* nominal, const float* expert, int Nn, int Ne,
                                       const int32_t* ep_offsets, const int32_t* row_episode, int n_ep,
                                       const icrl_cn_hyper_t* hp, const int32_t* perms, int batch_size, float* work,
                                       float* metrics, void* stream) {
  if (Nn <= 0 || Ne <= 0 || n_ep <= 0 || hp->iterations < 0 || batch_size <= 0 || perms == nullptr)
    return fail("icrl_cn_train_minibatch: needs nominal rows (%d), expert rows (%d), episodes (%d), batch_size (%d) > 0, iterations (%d) >= 0 "
                "and a permutation table", Nn, Ne, n_ep, batch_size, hp->iterations);
  CnTrainArgs a;
  size_t lds;
  if (int e = cn_dims_checked(cn, "icrl_cn_train_minibatch", &a.d, &lds)) return e;
  a.params = cn->params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.adam_t = adam_step;
  a.nominal = nominal; a.expert = expert; a.Nn = Nn; a.Ne = Ne; a.n_ep = n_ep;
  a.nb_n = (Nn + CN_ROWS - 1) / CN_ROWS; a.nb_e = (Ne + CN_ROWS - 1) / CN_ROWS;
  a.ep_off = ep_offsets; a.row_ep = row_episode; a.hp = *hp;
  size_t offs[9], total;
  cn_work_layout(cn->n_params, Nn, Ne, n_ep, offs, &total);
  a.start_preds = work + offs[0]; a.preds_n = work + offs[1]; a.preds_e = work + offs[2]; a.part = work + offs[3];
  a.ep_prod = work + offs[4]; a.ep_slog = work + offs[5]; a.normed = work + offs[6]; a.scal = work + offs[7];
  a.gpart = work + offs[8]; a.metrics = metrics; a.mb_idx = nullptr; a.mb_n = 0;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a.scal, 0, SC_COUNT * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  if (hp->iterations > 0) {
    e = hipMemsetAsync(metrics, 0, (size_t)hp->iterations * ICRL_CN_METRICS * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  const void* fns[3] = {(const void*)cn_forward_kernel, (const void*)cn_mb_forward_kernel, (const void*)cn_backward_kernel<true>};
  for (const void* f : fns) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int size = Nn < Ne ? Nn : Ne;                       // ref: constraint_net.py:304 — min(nom_size, exp_size)
  const int n_batches = (size + batch_size - 1) / batch_size;
  const int nb_all = a.nb_n + a.nb_e;
  for (int itr = 0; itr < hp->iterations; ++itr) {
    hipLaunchKernelGGL(cn_forward_kernel, dim3(nb_all), dim3(CN_TH), lds, s, a, itr);
    hipLaunchKernelGGL(cn_finalize_kernel, dim3(1), dim3(1024), 0, s, a, itr);
    for (int k = 0; k < n_batches; ++k) {
      CnTrainArgs b = a;
      b.mb_idx = perms + (size_t)itr * size + (size_t)k * batch_size;
      b.mb_n = (k + 1) * batch_size <= size ? batch_size : size - k * batch_size;
      b.nb_n = b.nb_e = (b.mb_n + CN_ROWS - 1) / CN_ROWS;
      const int nb = 2 * b.nb_n;
      hipLaunchKernelGGL(cn_mb_forward_kernel, dim3(nb), dim3(CN_TH), lds, s, b);
      hipLaunchKernelGGL(cn_mb_finalize_kernel, dim3(1), dim3(64), 0, s, b, itr);
      hipLaunchKernelGGL(cn_backward_kernel<true>, dim3(nb), dim3(CN_TH), lds, s, b, itr);
      hipLaunchKernelGGL(cn_adam_kernel, dim3((cn->n_params + 255) / 256), dim3(256), 0, s, b, itr, itr * n_batches + k);
    }
  }
  hipLaunchKernelGGL(cn_commit_kernel, dim3(1), dim3(64), 0, s, a);
  return (int)hipGetLastError();
}
