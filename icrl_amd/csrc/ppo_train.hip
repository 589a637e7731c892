// PPO-Lagrangian update as ONE persistent launch — gfx950.
//
// ref: stable_baselines3/ppo_lag/ppo_lag.py:196-299 (epoch / minibatch loop of PPOLagrangian.train),
//      common/buffers.py:594-627 (get / _get_samples), common/policies.py:752-767 (evaluate_actions),
//      common/distributions.py:143-171 (DiagGaussian log_prob / entropy), :274-288 (Categorical), torch.optim.Adam,
//      clip_grad_norm_.
//
// The reference performs n_epochs * ceil(T*N / batch) DEPENDENT optimiser steps on a 64..128-row minibatch through three
// small independent MLPs (pi, vf, cvf: obs -> 64 -> 64 -> {act | 1 | 1}).  Parity forbids re-ordering or merging those
// steps, so the step latency is what matters.  Mapping:
//
//   * grid = 3 workgroups x 512 threads: workgroup r owns network r.  The only quantity coupling the networks inside a
//     step is the global gradient norm (clip_grad_norm_ over ALL policy parameters): each workgroup publishes its partial
//     sum of squares as an 8-byte {step tag, value} granule (agent-scope relaxed store) and polls the other two (guide:
//     cdna_hip_programming.md §6 Guideline 16, form R2 — the datum is the flag).  Granule slots are double-buffered by step
//     parity; a workgroup can never be more than one step ahead of the others.  While the granules are in flight the
//     workgroup already stages the NEXT minibatch (LDS commit + advantage statistics).
//   * 8 waves = 2 per SIMD: wave (rt, hf) owns row tile rt (16 of the chunk's 64 rows) and column half hf of every 64-wide
//     GEMM output, so while one wave of a SIMD runs its VALU epilogue / LDS traffic the other one feeds the matrix core.
//   * the network's weights stay in LDS for the whole launch (fp32 master copy); Adam moments and the accumulating weight
//     gradients stay in REGISTERS in MFMA C-layout: the lane that receives dW[j][k] from the matrix core owns m, v and the
//     update of W[j][k].  Inside the loop only the gathered minibatch rows (prefetched one chunk ahead into registers,
//     their permutation indices two chunks ahead) and the 3 granules touch global memory.
//   * all eight GEMMs of a step (3 forward, 5 backward) run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).  K is
//     enumerated as k = 16*js + 4*(lane/16) + e: an operand whose K runs along the LDS row is fetched with ONE ds_read_b128
//     per four MFMA steps, conflict-free at row strides = 8 mod 16 floats; operands whose K runs across rows use
//     ds_read_b32.  Every GEMM first stages its operands in registers, then issues its MFMAs back to back.
//   * minibatches larger than 64 rows are processed in 64-row chunks that accumulate into the same gradient registers.
//
// Built with -ffp-contract=off; FMA is used only where written (fmaf / MFMA).
#include "ppo_common.h"

namespace icrl {

constexpr int TH = 512;  // threads per workgroup (8 waves, 2 per SIMD)
constexpr int SO = 24;   // LDS row stride of 16-wide matrices

template <int NT1>
struct Smem {  // offsets in floats (all multiples of 4: 16-byte aligned rows)
  static constexpr int O16 = 16 * NT1, SX = O16 + 8;
  static constexpr int W1 = 0;
  static constexpr int W2 = W1 + HD * SX;
  static constexpr int WH = W2 + HD * SH;
  static constexpr int B1 = WH + 16 * SH;
  static constexpr int B2 = B1 + HD;
  static constexpr int BH = B2 + HD;
  static constexpr int LS = BH + 16;
  static constexpr int X = LS + 16;
  static constexpr int H1 = X + RB * SX;
  static constexpr int H2 = H1 + RB * SH;     // h2, later dz1
  static constexpr int DZ = H2 + RB * SH;     // dz2; before the backward its rows hold the chunk's actions / scalars
  static constexpr int DO = DZ + RB * SH;     // head output, overwritten (after a barrier) by d loss / d output
  static constexpr int RED1 = DO + RB * SO;   // [4][64] per-row-tile column sums of dz1
  static constexpr int RED2 = RED1 + 4 * HD;  // [4][64] per-row-tile column sums of dz2
  static constexpr int PBH = RED2 + 4 * HD;   // [4][16] per-row-tile column sums of dOut
  static constexpr int PLS = PBH + 64;        // [4][16] per-row-tile d log_std partials
  static constexpr int PST = PLS + 64;        // [4][8] per-row-tile loss statistics
  static constexpr int MISC = PST + 32;       // [48] block-reduction scratch + broadcast scalars
  static constexpr int GAU = MISC + 48;       // [3][16] per-action 1/var, 0.5/var, log(sd) + log(sqrt(2 pi))
  static constexpr int TOTAL = GAU + 48;
  // per-row side data of the chunk lives in the (not yet used) dz rows: columns 0..15 actions, 16 old log-prob | old value,
  // 17 raw reward advantage | return, 18 raw cost advantage
  static constexpr int ACT = DZ, OLP = DZ + 16, ADR = DZ + 17, ADC = DZ + 18;
};

template <int NT1, bool DISC>
__global__ void __launch_bounds__(TH, 2) ppo_train_kernel(TrainArgs a) {
  using S = Smem<NT1>;
  constexpr int SX = S::SX;
  constexpr int NT1H = NT1 / 2;     // W1 / dW1 column tiles per wave
  constexpr int XR = (SX + 7) / 8;  // floats of an X row each of the 8 threads of a row stages
  static_assert(NT1 >= 2, "obs tiles are split between the two column halves");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int role = blockIdx.x;  // 0 policy, 1 reward critic, 2 cost critic
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = w & 3, hf = w >> 2;
  const int r = lane & 15, q = lane >> 4;
  const PolLayout& L = a.L;
  const int O = L.O, A = L.A;
  const int n_out = role == 0 ? A : 1;
  const int T = a.buf.T, N = a.buf.N;
  const int n_total = T * N;
  const int B = a.hp.batch_size;
  const int n_mb = (n_total + B - 1) / B;
  const int n_epochs = a.hp.n_epochs;
  const float nu = a.nu[0];

  // ---- global offsets of this role's tensors in the flat parameter buffer
  const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
  const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
  const int gbh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);

  // ---- load weights into LDS (zero padded), moments into registers (C-layout ownership)
  for (int i = tid; i < HD * SX; i += TH) { const int j = i / SX, k = i % SX; sm[S::W1 + i] = k < O ? a.params[gW1 + j * O + k] : 0.f; }
  for (int i = tid; i < HD * SH; i += TH) { const int j = i / SH, k = i % SH; sm[S::W2 + i] = k < HD ? a.params[gW2 + j * HD + k] : 0.f; }
  for (int i = tid; i < 16 * SH; i += TH) { const int o = i / SH, k = i % SH; sm[S::WH + i] = (o < n_out && k < HD) ? a.params[gWh + o * HD + k] : 0.f; }
  if (tid < HD) { sm[S::B1 + tid] = a.params[gb1 + tid]; sm[S::B2 + tid] = a.params[gb2 + tid]; }
  if (tid < 16) { sm[S::BH + tid] = tid < n_out ? a.params[gbh + tid] : 0.f; sm[S::LS + tid] = (!DISC && role == 0 && tid < A) ? a.params[L.log_std + tid] : 0.f; }

  // wave (rt, hf) owns: W1 rows 16rt.. x column tiles hf*NT1H..; W2 rows 16rt.. x column tiles 2hf, 2hf+1;
  // (hf == 0 only) head-weight columns 16rt..16rt+15
  f32x4 mW1[NT1H], vW1[NT1H], gW1r[NT1H], mW2[2], vW2[2], gW2r[2], mWh, vWh, gWhr;
#pragma unroll
  for (int c = 0; c < NT1H; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * rt + 4 * q + i, k = 16 * (hf * NT1H + c) + r;
      mW1[c][i] = k < O ? a.exp_avg[gW1 + j * O + k] : 0.f;
      vW1[c][i] = k < O ? a.exp_avg_sq[gW1 + j * O + k] : 0.f;
    }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * rt + 4 * q + i, k = 16 * (2 * hf + c) + r;
      mW2[c][i] = a.exp_avg[gW2 + j * HD + k];
      vW2[c][i] = a.exp_avg_sq[gW2 + j * HD + k];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = 4 * q + i, j = 16 * rt + r;
    const bool own = hf == 0 && o < n_out;
    mWh[i] = own ? a.exp_avg[gWh + o * HD + j] : 0.f;
    vWh[i] = own ? a.exp_avg_sq[gWh + o * HD + j] : 0.f;
  }
  // thread-owned vector parameters: tid 0..63 b1, 64..127 b2, 128..143 head bias, 144..159 log_std (policy only)
  int vec_g = -1, vec_s = 0;
  if (tid < 64) { vec_g = gb1 + tid; vec_s = S::B1 + tid; }
  else if (tid < 128) { vec_g = gb2 + tid - 64; vec_s = S::B2 + tid - 64; }
  else if (tid < 144) { if (tid - 128 < n_out) { vec_g = gbh + tid - 128; vec_s = S::BH + tid - 128; } }
  else if (tid < 160) { if (!DISC && role == 0 && tid - 144 < A) { vec_g = L.log_std + tid - 144; vec_s = S::LS + tid - 144; } }
  float mB = 0.f, vB = 0.f, gB = 0.f;
  if (vec_g >= 0) { mB = a.exp_avg[vec_g]; vB = a.exp_avg_sq[vec_g]; }

  const int t0 = a.adam_t[0];
  double b1pow = pow((double)a.hp.adam_beta1, (double)t0), b2pow = pow((double)a.hp.adam_beta2, (double)t0);
  const float w1 = (float)(1.0 - (double)a.hp.adam_beta1);
  const float w2 = (float)(1.0 - (double)a.hp.adam_beta2);
  const float clip = a.hp.clip_range;
  const float vclip = role == 1 ? a.hp.clip_range_reward_vf : a.hp.clip_range_cost_vf;
  const float vcoef = role == 1 ? a.hp.reward_vf_coef : a.hp.cost_vf_coef;

  // ---------------------------------------------------------------------------------------------------------------
  // row stream helpers
  // ---------------------------------------------------------------------------------------------------------------
  auto cur_rows = [&](const Cursor& c) {   // rows of the 64-row chunk that starts at c
    int left = B - c.m;
    if (n_total - c.p < left) left = n_total - c.p;
    return left < RB ? left : RB;
  };
  auto cur_advance = [&](Cursor c) {
    const int rows = cur_rows(c);
    c.p += rows; c.m += rows;
    if (c.p >= n_total) { c.p = 0; c.m = 0; ++c.e; }
    else if (c.m >= B) c.m = 0;
    return c;
  };
  // flat env-major index -> [T,N] storage offset (ref: buffers.py:53-65): env = idx / T, t = idx % T
  auto to_off = [&](int idx) -> unsigned {
    unsigned env = __umulhi((unsigned)idx, a.t_magic);
    int t = idx - (int)env * T;
    if (t >= T) { t -= T; ++env; }
    if (t >= T) { t -= T; ++env; }
    return (unsigned)t * (unsigned)N + env;
  };
  // this thread's row of a chunk: b = tid/8 (part = tid%8 splits the row's floats)
  const int gb_row = tid >> 3, gpart = tid & 7;
  auto load_idx = [&](const Cursor& c) -> int {
    if (c.e >= n_epochs || gb_row >= cur_rows(c)) return -1;
    return a.perms[(size_t)c.e * n_total + c.p + gb_row];
  };
  // row data of one chunk held in registers between "issue" and "commit"
  float px[XR], pact[2], psc0 = 0.f, psc1 = 0.f, psc2 = 0.f;
  auto issue_rows = [&](int idx) {
    const bool valid = idx >= 0;
    const size_t off = valid ? (size_t)to_off(idx) : 0;
    const float* orow = a.buf.observations + off * O;
#pragma unroll
    for (int i = 0; i < XR; ++i) { const int k = gpart + 8 * i; px[i] = (valid && k < O) ? orow[k] : 0.f; }
    if (role == 0) {
      const float* arow = a.buf.actions + off * a.buf.act_store;
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int k = gpart + 8 * i; pact[i] = (valid && k < a.buf.act_store) ? arow[k] : 0.f; }
      if (gpart == 0) {
        psc0 = valid ? a.buf.log_probs[off] : 0.f;
        psc1 = valid ? a.buf.reward_advantages[off] : 0.f;
        psc2 = valid ? a.buf.cost_advantages[off] : 0.f;
      }
    } else if (gpart == 0) {
      const float* rets = role == 1 ? a.buf.reward_returns : a.buf.cost_returns;
      const float* olds = role == 1 ? a.buf.reward_values : a.buf.cost_values;
      psc1 = valid ? rets[off] : 0.f;
      psc0 = valid ? olds[off] : 0.f;
    }
  };
  auto commit_rows = [&]() {
#pragma unroll
    for (int i = 0; i < XR; ++i) { const int k = gpart + 8 * i; if (k < SX) sm[S::X + gb_row * SX + k] = px[i]; }
    if (role == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sm[S::ACT + gb_row * SH + gpart + 8 * i] = pact[i];
      if (gpart == 0) { sm[S::OLP + gb_row * SH] = psc0; sm[S::ADR + gb_row * SH] = psc1; sm[S::ADC + gb_row * SH] = psc2; }
    } else if (gpart == 0) {
      sm[S::OLP + gb_row * SH] = psc0; sm[S::ADR + gb_row * SH] = psc1;
    }
  };
  // advantage statistics of a minibatch (policy role): thread tid < nb holds row tid's (A_r, A_c)
  float sar = 0.f, sac = 0.f;
  auto mb_rows_at = [&](int p) { const int left = n_total - p; return left < B ? left : B; };
  auto stat_idx = [&](int e, int p) -> int {   // minibatch starting at position p of epoch e
    if (role != 0 || e >= n_epochs || tid >= mb_rows_at(p)) return -1;
    return a.perms[(size_t)e * n_total + p + tid];
  };
  auto issue_stats = [&](int idx) {
    sar = 0.f; sac = 0.f;
    if (idx >= 0) {
      const unsigned off = to_off(idx);
      sar = a.buf.reward_advantages[off];
      sac = a.buf.cost_advantages[off];
    }
  };
  float mean_r = 0.f, std_r = 1.f, mean_c = 0.f;
  auto compute_stats = [&](int nb) {  // all threads of a policy workgroup; uses sar / sac.  ONE block reduction.
    if (role != 0) return;
    const bool in = tid < nb;
    float s_r = wave_sum_fast(in ? sar : 0.f), s_c = wave_sum_fast(in ? sac : 0.f), s_rr = wave_sum_fast(in ? sar * sar : 0.f);
    lds_barrier();
    if (lane == 0) { sm[S::MISC + 16 + w] = s_r; sm[S::MISC + 24 + w] = s_c; sm[S::MISC + 32 + w] = s_rr; }
    lds_barrier();
    // minibatches have <= 128 rows: only waves 0 and 1 carry data
    s_r = sm[S::MISC + 16] + sm[S::MISC + 17];
    s_c = sm[S::MISC + 24] + sm[S::MISC + 25];
    s_rr = sm[S::MISC + 32] + sm[S::MISC + 33];
    mean_r = s_r / (float)nb;
    mean_c = s_c / (float)nb;
    // unbiased variance from the raw moments (advantages are O(1): fp32 cancellation stays ~1e-6 relative)
    const float var = fmaxf(s_rr - s_r * mean_r, 0.f) / (float)(nb - 1);
    std_r = sqrtf(var);
  };
  auto refresh_gauss = [&]() {   // threads 144..159 own log_std: derived constants of the Gaussian head
    if (!DISC && role == 0 && tid >= 144 && tid < 160) {
      const int k = tid - 144;
      const float sd = __expf(sm[S::LS + k]);
      const float iv = __builtin_amdgcn_rcpf(sd * sd);
      sm[S::GAU + k] = k < A ? iv : 0.f;
      sm[S::GAU + 16 + k] = k < A ? 0.5f * iv : 0.f;
      sm[S::GAU + 32 + k] = k < A ? __logf(sd) + LOG_SQRT_2PI_F : 0.f;
    }
  };
  // running statistics (thread 0 of each role)
  float st_ent = 0.f, st_pg = 0.f, st_vl = 0.f, st_cf = 0.f, last_loss = 0.f;
  int steps_done = 0, early_stop_epoch = n_epochs, status = 0;
  if (tid == 0) { sm[S::MISC + 12] = 0.f; sm[S::MISC + 13] = 0.f; }

  // ---- pipeline prologue: chunk 0 rows -> LDS, chunk 1 / 2 indices in flight; minibatch 0 statistics
  Cursor c_nx2 = cur_advance(Cursor{0, 0, 0});
  int idx_next = load_idx(c_nx2);
  c_nx2 = cur_advance(c_nx2);
  int idx_nx2 = load_idx(c_nx2);
  issue_rows(load_idx(Cursor{0, 0, 0}));
  issue_stats(stat_idx(0, 0));
  refresh_gauss();
  // (epoch, position) of the minibatch whose statistics indices are in flight: the one after the current
  int s_e = 0, s_p = mb_rows_at(0);
  if (s_p >= n_total) { s_p = 0; s_e = 1; }
  int sidx_next = stat_idx(s_e, s_p);
  __syncthreads();
  commit_rows();
  compute_stats(mb_rows_at(0));
  __syncthreads();

  const bool prof = (a.hp._pad & 1) != 0;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = prof ? stamp() : 0ull;

  unsigned step = 0;
  bool stop = false;
  for (int epoch = 0; epoch < n_epochs && !stop; ++epoch) {
    float kl_sum = 0.f;  // thread 0, policy role
    for (int mb = 0; mb < n_mb && !stop; ++mb) {
      ++step;
      const int nb = mb_rows_at(mb * B);
      const float c_mean_r = mean_r, c_mean_c = mean_c;   // statistics of THIS minibatch
      const float c_istd_r = 1.f / (std_r + 1e-8f);
      const float cpol_nb = 1.f / ((1.f + nu) * (float)nb);
      // statistics prefetch: advantages of the NEXT minibatch's rows (indices loaded a step ago), indices of the one after
      issue_stats(sidx_next);
      {
        s_p += mb_rows_at(s_p);
        if (s_p >= n_total) { s_p = 0; ++s_e; }
        sidx_next = stat_idx(s_e, s_p);
      }
      // ---- zero gradient accumulators
#pragma unroll
      for (int c = 0; c < NT1H; ++c) gW1r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) gW2r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      gWhr = f32x4{0.f, 0.f, 0.f, 0.f};
      gB = 0.f;
      float mb_s0 = 0.f, mb_s1 = 0.f, mb_s2 = 0.f, mb_s3 = 0.f, mb_s4 = 0.f;  // thread 0: minibatch sums of the loss statistics

      const int n_chunks = (nb + RB - 1) / RB;
      for (int ch = 0; ch < n_chunks; ++ch) {
        const int nrows = (nb - ch * RB) < RB ? (nb - ch * RB) : RB;
        if (ch > 0) {  // chunk 0 of a step was committed during the previous step's granule wait (or the prologue)
          commit_rows();
          lds_barrier();
        }
        // ================= forward: wave (rt, hf): rows 16rt.., output columns 32hf..32hf+31 =================
        f32x4 h1t[2], h2t[2];   // this wave's h1 / h2 tiles stay in registers for the backward epilogues
        {
          f32x4 av[NT1], bv[2][NT1];
          const float* pa = sm + S::X + (16 * rt + r) * SX + 4 * q;
          const float* pb = sm + S::W1 + (32 * hf + r) * SX + 4 * q;
#pragma unroll
          for (int js = 0; js < NT1; ++js) {
            av[js] = lds128(pa + 16 * js);
#pragma unroll
            for (int c = 0; c < 2; ++c) bv[c][js] = lds128(pb + c * 16 * SX + 16 * js);
          }
          __builtin_amdgcn_sched_barrier(0);   // all operand reads are issued before the first MFMA
          f32x4 acc[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int js = 0; js < NT1; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA_F32(av[js][e], bv[c][js][e], acc[c]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = 32 * hf + 16 * c + r;
            const float bias = sm[S::B1 + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h1t[c][i] = fast_tanh(acc[c][i] + bias);
              sm[S::H1 + (16 * rt + 4 * q + i) * SH + col] = h1t[c][i];
            }
          }
        }
        // prefetch (under the first GEMM's shadow): rows of the next chunk of the stream, indices of the one after the next
        issue_rows(idx_next);
        idx_next = idx_nx2;
        c_nx2 = cur_advance(c_nx2);
        idx_nx2 = load_idx(c_nx2);
        lds_barrier();  // (A1) both column halves of h1 written
        {
          f32x4 av[4], bv[2][4];
          const float* pa = sm + S::H1 + (16 * rt + r) * SH + 4 * q;
          const float* pb = sm + S::W2 + (32 * hf + r) * SH + 4 * q;
#pragma unroll
          for (int js = 0; js < 4; ++js) {
            av[js] = lds128(pa + 16 * js);
#pragma unroll
            for (int c = 0; c < 2; ++c) bv[c][js] = lds128(pb + c * 16 * SH + 16 * js);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA_F32(av[js][e], bv[c][js][e], acc[c]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = 32 * hf + 16 * c + r;
            const float bias = sm[S::B2 + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h2t[c][i] = fast_tanh(acc[c][i] + bias);
              sm[S::H2 + (16 * rt + 4 * q + i) * SH + col] = h2t[c][i];
            }
          }
        }
        lds_barrier();  // (A2) h2 complete
        if (hf == 0) {    // head: [16 rows] x [16 outputs], K = 64
          f32x4 av[4], bv[4];
          const float* pa = sm + S::H2 + (16 * rt + r) * SH + 4 * q;
          const float* pb = sm + S::WH + r * SH + 4 * q;
#pragma unroll
          for (int js = 0; js < 4; ++js) { av[js] = lds128(pa + 16 * js); bv[js] = lds128(pb + 16 * js); }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc[4];   // four independent chains (one per js), summed afterwards
#pragma unroll
          for (int js = 0; js < 4; ++js) {
            acc[js] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[js] = MFMA_F32(av[js][e], bv[js][e], acc[js]);
          }
          const float bias = sm[S::BH + r];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            sm[S::DO + (16 * rt + 4 * q + i) * SO + r] = ((acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i])) + bias;
        }
        lds_barrier();  // (A3) head outputs visible to both waves of a row tile
        STAMP(0)   // forward
        // ============ loss + d loss / d head output: row 16rt + r; lane group q and half hf split the actions ============
        {
          const int b = 16 * rt + r;
          const bool valid = b < nrows;
          float* dor = sm + S::DO + b * SO;     // holds the head output of row b; overwritten with its gradient
          if (role == 0 && DISC) {
            // Categorical(logits) (ref: distributions.py:274-288): lane group q holds the logits k = q + 4 i of row b
            float lg[4], pr[4], zmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = q + 4 * i;
              lg[i] = k < A ? dor[k] : -INFINITY;
              zmax = fmaxf(zmax, lg[i]);
            }
            zmax = xor16_max(xor32_max(zmax));
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) se += (q + 4 * i < A) ? expf(lg[i] - zmax) : 0.f;
            se = quad_rows_sum(se);
            const float lse = zmax + logf(se);
            const int act = (int)sm[S::ACT + b * SH];
            float lp = 0.f, ent = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = q + 4 * i;
              lg[i] = k < A ? lg[i] - lse : 0.f;         // log-softmax
              pr[i] = k < A ? expf(lg[i]) : 0.f;
              lp += (k == act) ? lg[i] : 0.f;
              ent -= pr[i] * lg[i];
            }
            lp = quad_rows_sum(lp);
            ent = quad_rows_sum(ent);
            const float old_lp = sm[S::OLP + b * SH];
            const float ratio = __expf(lp - old_lp);
            const float Ar = (sm[S::ADR + b * SH] - c_mean_r) * c_istd_r;
            const float Ac = sm[S::ADC + b * SH] - c_mean_c;
            const float s1 = Ar * ratio;
            const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
            const float s2 = Ar * rc;
            const float gsel = (s1 <= s2) ? Ar : 0.f;
            const float dlp = valid ? cpol_nb * (-gsel + nu * Ac) * ratio : 0.f;
            // loss += ent_coef * (-mean H):  d/dz_k = ent_coef / nb * p_k (log p_k + H)
            const float dent = valid ? a.hp.ent_coef / (float)nb : 0.f;
            lds_barrier();   // (L)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
              const int k = q + 4 * (2 * hf + ii);
              const float pk = hf == 0 ? pr[ii] : pr[2 + ii];
              const float lk = hf == 0 ? lg[ii] : lg[2 + ii];
              const float dk = k < A ? dlp * ((k == act ? 1.f : 0.f) - pk) + dent * (pk * (lk + ent)) : 0.f;
              dor[k] = dk;
              const float colsum = sum16(dk);
              if (r == 0) { sm[S::PBH + rt * 16 + k] = colsum; sm[S::PLS + rt * 16 + k] = 0.f; }
            }
            if (hf == 0) {
              const float v0 = sum16(valid ? fminf(s1, s2) : 0.f);
              const float v1 = sum16(valid ? Ac * ratio : 0.f);
              const float v2 = sum16(valid ? (fabsf(ratio - 1.f) > clip ? 1.f : 0.f) : 0.f);
              const float v3 = sum16(valid ? old_lp - lp : 0.f);
              const float v4 = sum16(valid ? ent : 0.f);
              if (lane == 0) { sm[S::PST + rt * 8 + 0] = v0; sm[S::PST + rt * 8 + 1] = v1; sm[S::PST + rt * 8 + 2] = v2; sm[S::PST + rt * 8 + 3] = v3; sm[S::PST + rt * 8 + 4] = v4; }
            }
          } else if (role == 0) {
            float dd[4], iv[4];
            float lp = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = q + 4 * i;                      // every wave of the pair evaluates the whole row
              dd[i] = sm[S::ACT + b * SH + k] - dor[k];     // pad actions / outputs are 0
              iv[i] = sm[S::GAU + k];
              lp += -(dd[i] * dd[i]) * sm[S::GAU + 16 + k] - sm[S::GAU + 32 + k];
            }
            lp = quad_rows_sum(lp);
            const float old_lp = sm[S::OLP + b * SH];
            const float ratio = __expf(lp - old_lp);
            const float Ar = (sm[S::ADR + b * SH] - c_mean_r) * c_istd_r;
            const float Ac = sm[S::ADC + b * SH] - c_mean_c;
            const float s1 = Ar * ratio;
            const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
            const float s2 = Ar * rc;
            const float gsel = (s1 <= s2) ? Ar : 0.f;                       // d min(s1, s2) / d ratio
            const float dlp = valid ? cpol_nb * (-gsel + nu * Ac) * ratio : 0.f;  // d loss / d log_prob
            lds_barrier();   // (L) both waves of the pair have read the head outputs before they are overwritten
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
              const int i = 2 * hf + ii;                    // this wave writes / reduces the actions k = q + 4i of its half
              const int k = q + 4 * i;
              const float ddi = hf == 0 ? dd[ii] : dd[2 + ii];
              const float ivi = hf == 0 ? iv[ii] : iv[2 + ii];
              const float dk = dlp * (ddi * ivi);
              const float lk = (k < A) ? dlp * ((ddi * ddi) * ivi - 1.f) : 0.f;
              dor[k] = dk;
              const float colsum = sum16(dk);
              const float lssum = sum16(lk);
              if (r == 0) { sm[S::PBH + rt * 16 + k] = colsum; sm[S::PLS + rt * 16 + k] = lssum; }
            }
            if (hf == 0) {
              const float v0 = sum16(valid ? fminf(s1, s2) : 0.f);
              const float v1 = sum16(valid ? Ac * ratio : 0.f);
              const float v2 = sum16(valid ? (fabsf(ratio - 1.f) > clip ? 1.f : 0.f) : 0.f);
              const float v3 = sum16(valid ? old_lp - lp : 0.f);
              if (lane == 0) { sm[S::PST + rt * 8 + 0] = v0; sm[S::PST + rt * 8 + 1] = v1; sm[S::PST + rt * 8 + 2] = v2; sm[S::PST + rt * 8 + 3] = v3; }
            }
          } else {
            const float v = dor[0];
            const float R = sm[S::ADR + b * SH];
            float vp = v, pass = 1.f;
            if (vclip >= 0.f) {
              const float old = sm[S::OLP + b * SH];
              const float dv = v - old;
              vp = old + fminf(fmaxf(dv, -vclip), vclip);
              pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
            }
            const float e = vp - R;
            const float d0 = valid ? vcoef * 2.f * e / (float)nb * pass : 0.f;
            lds_barrier();   // (L)
            if (hf == 0) {
              if (q == 0) {
                dor[0] = d0;
#pragma unroll
                for (int k = 1; k < 16; ++k) dor[k] = 0.f;
              }
              const float colsum = sum16(d0);
              const float se = sum16(valid ? e * e : 0.f);
              if (lane == 0) {
                sm[S::PBH + rt * 16] = colsum;
#pragma unroll
                for (int k = 1; k < 16; ++k) sm[S::PBH + rt * 16 + k] = 0.f;
                sm[S::PST + rt * 8 + 0] = se;
              }
            }
          }
        }
        lds_barrier();  // (A4) d loss / d output of the row tile complete (written by both halves)
        STAMP(1)   // loss
        // ================= backward =================
        {  // dH2 = dOut . Wh  -> dz2 = dH2 * (1 - h2^2) (own rows, own column half), column sums for d b2
          const f32x4 av = lds128(sm + S::DO + (16 * rt + r) * SO + 4 * q);   // k = a = 4q + e
          float bv[2][4];
          const float* pb = sm + S::WH + (4 * q) * SH + 32 * hf + r;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c) bv[c][e] = pb[e * SH + 16 * c];
          f32x4 acc[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = MFMA_F32(av[e], bv[c][e], acc[c]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = 32 * hf + 16 * c + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc[c][i] = acc[c][i] * (1.f - h2t[c][i] * h2t[c][i]);
              sm[S::DZ + (16 * rt + 4 * q + i) * SH + col] = acc[c][i];
            }
            const float cs = tile_colsum(acc[c]);
            if (q == 0) sm[S::RED2 + rt * HD + col] = cs;
          }
        }
        lds_barrier();  // (2) dz2, dOut, h2 of ALL rows visible
        if (hf == 0) {  // dWh[o][j] += sum_b dOut[b][o] h2[b][j]   (columns 16rt..16rt+15); b = 16 js + 4 q + e
          float av[4][4], bv[4][4];
          const float* pa = sm + S::DO + (4 * q) * SO + r;
          const float* pb = sm + S::H2 + (4 * q) * SH + 16 * rt + r;
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) { av[js][e] = pa[(16 * js + e) * SO]; bv[js][e] = pb[(16 * js + e) * SH]; }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc[4];
#pragma unroll
          for (int js = 0; js < 4; ++js) {
            acc[js] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[js] = MFMA_F32(av[js][e], bv[js][e], acc[js]);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) gWhr[i] += (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
        }
        // owners fold the per-row-tile partials of this chunk (b2, head bias, log_std, loss statistics)
        if (tid >= 64 && tid < 128) {
          const int j = tid - 64;
          gB += (sm[S::RED2 + j] + sm[S::RED2 + HD + j]) + (sm[S::RED2 + 2 * HD + j] + sm[S::RED2 + 3 * HD + j]);
        } else if (tid >= 128 && tid < 144) {
          const int k = tid - 128;
          gB += (sm[S::PBH + k] + sm[S::PBH + 16 + k]) + (sm[S::PBH + 32 + k] + sm[S::PBH + 48 + k]);
        } else if (tid >= 144 && tid < 160) {
          const int k = tid - 144;
          gB += (sm[S::PLS + k] + sm[S::PLS + 16 + k]) + (sm[S::PLS + 32 + k] + sm[S::PLS + 48 + k]);
        }
        if (tid == 0) {
          mb_s0 += (sm[S::PST + 0] + sm[S::PST + 8]) + (sm[S::PST + 16] + sm[S::PST + 24]);
          mb_s1 += (sm[S::PST + 1] + sm[S::PST + 9]) + (sm[S::PST + 17] + sm[S::PST + 25]);
          mb_s2 += (sm[S::PST + 2] + sm[S::PST + 10]) + (sm[S::PST + 18] + sm[S::PST + 26]);
          mb_s3 += (sm[S::PST + 3] + sm[S::PST + 11]) + (sm[S::PST + 19] + sm[S::PST + 27]);
          if (DISC) mb_s4 += (sm[S::PST + 4] + sm[S::PST + 12]) + (sm[S::PST + 20] + sm[S::PST + 28]);
        }
        lds_barrier();  // (2b) every wave is done reading h2: its buffer becomes dz1
        {  // dH1 = dz2 . W2 (own rows, own column half) -> dz1 = dH1 * (1 - h1^2), stored over h2
          f32x4 av[4];
          const float* pa = sm + S::DZ + (16 * rt + r) * SH + 4 * q;
          const float* pb = sm + S::W2 + (4 * q) * SH + 32 * hf + r;
#pragma unroll
          for (int js = 0; js < 4; ++js) av[js] = lds128(pa + 16 * js);
          float bv[4][2][4];
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) bv[js][c][e] = pb[(16 * js + e) * SH + 16 * c];
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) acc[c] = MFMA_F32(av[js][e], bv[js][c][e], acc[c]);
          // dW2[j][k] += sum_b dz2[b][j] h1[b][k]  (rows j = 16rt.., columns 32hf..): its MFMAs are independent of the
          // epilogue below, so the scheduler can hide the epilogue's VALU / LDS work under them
          float a2[4][4], b2v[4][2][4];
          const float* pa2 = sm + S::DZ + (4 * q) * SH + 16 * rt + r;
          const float* pb2 = sm + S::H1 + (4 * q) * SH + 32 * hf + r;
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a2[js][e] = pa2[(16 * js + e) * SH];
#pragma unroll
              for (int c = 0; c < 2; ++c) b2v[js][c][e] = pb2[(16 * js + e) * SH + 16 * c];
            }
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 2; ++c) gW2r[c] = MFMA_F32(a2[js][e], b2v[js][c][e], gW2r[c]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = 32 * hf + 16 * c + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc[c][i] = acc[c][i] * (1.f - h1t[c][i] * h1t[c][i]);
              sm[S::H2 + (16 * rt + 4 * q + i) * SH + col] = acc[c][i];
            }
            const float cs = tile_colsum(acc[c]);
            if (q == 0) sm[S::RED1 + rt * HD + col] = cs;
          }
        }
        lds_barrier();  // (3) dz1 of all rows visible
        {  // dW1[j][k] += sum_b dz1[b][j] x[b][k]   (rows j = 16rt.., column tiles hf*NT1H..)
          float av[4][4];
          const float* pa = sm + S::H2 + (4 * q) * SH + 16 * rt + r;
          const float* pb = sm + S::X + (4 * q) * SX + 16 * hf * NT1H + r;
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) av[js][e] = pa[(16 * js + e) * SH];
#pragma unroll
          for (int js = 0; js < 4; ++js) {
            float bv[NT1H][4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < NT1H; ++c) bv[c][e] = pb[(16 * js + e) * SX + 16 * c];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < NT1H; ++c) gW1r[c] = MFMA_F32(av[js][e], bv[c][e], gW1r[c]);
          }
        }
        if (tid < 64) gB += (sm[S::RED1 + tid] + sm[S::RED1 + HD + tid]) + (sm[S::RED1 + 2 * HD + tid] + sm[S::RED1 + 3 * HD + tid]);
        lds_barrier();  // (4) chunk buffers free
        STAMP(2)   // backward
      }  // chunks

      // entropy term of the policy loss: d(ent_coef * -mean(H)) / d log_std = -ent_coef (H = sum_a 0.5 + 0.5 log 2pi + log sigma)
      if (!DISC && role == 0 && tid >= 144 && tid < 144 + A) gB += -a.hp.ent_coef;

      // ================= global gradient norm: local sum of squares -> 8-byte granules =================
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < NT1H; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += gW1r[c][i] * gW1r[c][i];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += gW2r[c][i] * gW2r[c][i];
#pragma unroll
      for (int i = 0; i < 4; ++i) ss += gWhr[i] * gWhr[i];
      if (vec_g >= 0) ss += gB * gB;
      ss = wave_sum_fast(ss);
      if (lane == 0) sm[S::MISC + w] = ss;
      lds_barrier();
      // the early-stop decision rides on the policy workgroup's granule; publish first, book-keep afterwards
      if (tid == 0) {
        ss = ((sm[S::MISC + 0] + sm[S::MISC + 1]) + (sm[S::MISC + 2] + sm[S::MISC + 3])) +
             ((sm[S::MISC + 4] + sm[S::MISC + 5]) + (sm[S::MISC + 6] + sm[S::MISC + 7]));
        bool want_stop = false;
        float mean_kl = 0.f;
        if (role == 0) {
          kl_sum += mb_s3 / (float)nb;
          if (mb == n_mb - 1) {
            mean_kl = kl_sum / (float)n_mb;
            if (a.hp.use_target_kl && mean_kl > 1.5f * a.hp.target_kl) { want_stop = true; early_stop_epoch = epoch; }
          }
        }
        const unsigned tag = step | (want_stop ? 0x80000000u : 0u);
        __hip_atomic_store(a.xch + (step & 1) * 4 + role, ((u64)tag << 32) | (u64)__float_as_uint(ss), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        ++steps_done;
        if (role == 0) {
          float ent = 0.f;
          if (DISC) ent = mb_s4 / (float)nb;
          else for (int k = 0; k < A; ++k) ent += HALF_LOG_2PI_PLUS_HALF_F + sm[S::LS + k];
          const float entropy_loss = -ent;
          const float pl = (-(mb_s0 / (float)nb) + nu * (mb_s1 / (float)nb)) / (1.f + nu);
          st_ent += entropy_loss; st_pg += pl; st_cf += mb_s2 / (float)nb;
          last_loss = pl + a.hp.ent_coef * entropy_loss;
          if (mb == n_mb - 1) { a.stats[32 + epoch] = mean_kl; a.stats[7] = mean_kl; }
        } else {
          const float vl = mb_s0 / (float)nb;
          st_vl += vl;
          last_loss = vl;
        }
      }
      STAMP(3)   // gradient norm + publish
      // ---- while the granules travel: stage the next minibatch (rows -> LDS, advantage statistics)
      commit_rows();
      {
        const int pn = (mb + 1 < n_mb) ? (mb + 1) * B : 0;
        compute_stats(mb_rows_at(pn));
      }
      STAMP(4)   // next-minibatch staging
      if (tid < 3) {
        u64 v = 0;
        int spins = 0;
        bool ok = false;
        while (spins < (1 << 24)) {
          v = __hip_atomic_load(a.xch + (step & 1) * 4 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)((v >> 32) & 0x7fffffffu) == step) { ok = true; break; }
          __builtin_amdgcn_s_sleep(1);
          ++spins;
        }
        sm[S::MISC + 8 + tid] = __uint_as_float((unsigned)(v & 0xffffffffu));
        if (tid == 0) sm[S::MISC + 12] = (v >> 63) ? 1.f : 0.f;
        if (!ok) sm[S::MISC + 13] = 1.f;
      } else if (tid == 3) {   // Adam's bias corrections in double, once per step, while the others poll
        b1pow *= (double)a.hp.adam_beta1;
        b2pow *= (double)a.hp.adam_beta2;
        sm[S::MISC + 14] = (float)((double)a.hp.lr / (1.0 - b1pow));
        sm[S::MISC + 15] = (float)(1.0 / sqrt(1.0 - b2pow));
      }
      lds_barrier();
      STAMP(5)   // granule wait
      const float total = sqrtf((sm[S::MISC + 8] + sm[S::MISC + 9]) + sm[S::MISC + 10]);
      stop = sm[S::MISC + 12] != 0.f;
      if (sm[S::MISC + 13] != 0.f) { status = 1; stop = true; }
      float coef = a.hp.max_grad_norm / (total + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;

      // ================= Adam (torch.optim.Adam, single-tensor form) on register-resident moments =================
      const float step_size = sm[S::MISC + 14];
      const float inv_bc2_sqrt = sm[S::MISC + 15];
      const float b2f = a.hp.adam_beta2, epsf = a.hp.adam_eps;
      auto adam = [&](float g, float& m, float& v, float p) -> float {
        g = g * coef;
        m = m + (g - m) * w1;
        v = v * b2f + w2 * (g * g);
        const float denom = __builtin_amdgcn_sqrtf(v) * inv_bc2_sqrt + epsf;     // v_sqrt_f32 / v_rcp_f32: 1 ulp each
        return p - step_size * (m * __builtin_amdgcn_rcpf(denom));
      };
      if (status == 0) {
#pragma unroll
        for (int c = 0; c < NT1H; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int j = 16 * rt + 4 * q + i, k = 16 * (hf * NT1H + c) + r;
            if (k < O) { float* pw = sm + S::W1 + j * SX + k; float m_ = mW1[c][i], v_ = vW1[c][i]; *pw = adam(gW1r[c][i], m_, v_, *pw); mW1[c][i] = m_; vW1[c][i] = v_; }
          }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float* pw = sm + S::W2 + (16 * rt + 4 * q + i) * SH + 16 * (2 * hf + c) + r;
            float m_ = mW2[c][i], v_ = vW2[c][i];
            *pw = adam(gW2r[c][i], m_, v_, *pw);
            mW2[c][i] = m_; vW2[c][i] = v_;
          }
        if (hf == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int o = 4 * q + i;
            if (o < n_out) { float* pw = sm + S::WH + o * SH + 16 * rt + r; float m_ = mWh[i], v_ = vWh[i]; *pw = adam(gWhr[i], m_, v_, *pw); mWh[i] = m_; vWh[i] = v_; }
          }
        }
        if (vec_g >= 0) sm[vec_s] = adam(gB, mB, vB, sm[vec_s]);
        refresh_gauss();
      }
      if (tid == 0) { sm[S::MISC + 12] = 0.f; sm[S::MISC + 13] = 0.f; }
      lds_barrier();
      STAMP(6)   // Adam
    }  // minibatches
  }    // epochs

  __syncthreads();
  // ---- write back weights, moments, statistics
  for (int i = tid; i < HD * O; i += TH) { const int j = i / O, k = i % O; a.params[gW1 + i] = sm[S::W1 + j * SX + k]; }
  for (int i = tid; i < HD * HD; i += TH) { const int j = i / HD, k = i % HD; a.params[gW2 + i] = sm[S::W2 + j * SH + k]; }
  for (int i = tid; i < n_out * HD; i += TH) { const int o = i / HD, k = i % HD; a.params[gWh + i] = sm[S::WH + o * SH + k]; }
  if (vec_g >= 0) { a.params[vec_g] = sm[vec_s]; a.exp_avg[vec_g] = mB; a.exp_avg_sq[vec_g] = vB; }
#pragma unroll
  for (int c = 0; c < NT1H; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * rt + 4 * q + i, k = 16 * (hf * NT1H + c) + r;
      if (k < O) { a.exp_avg[gW1 + j * O + k] = mW1[c][i]; a.exp_avg_sq[gW1 + j * O + k] = vW1[c][i]; }
    }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * rt + 4 * q + i, k = 16 * (2 * hf + c) + r;
      a.exp_avg[gW2 + j * HD + k] = mW2[c][i];
      a.exp_avg_sq[gW2 + j * HD + k] = vW2[c][i];
    }
  if (hf == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = 4 * q + i, j = 16 * rt + r;
      if (o < n_out) { a.exp_avg[gWh + o * HD + j] = mWh[i]; a.exp_avg_sq[gWh + o * HD + j] = vWh[i]; }
    }
  }
  if (tid == 0 && prof) {
    for (int k = 0; k < 7; ++k) {
      const int slot = 12 + 7 * role + k;
      if (slot < 32) a.stats[slot] = (float)((double)ph[k] / (double)(steps_done > 0 ? steps_done : 1));
    }
  }
  if (tid == 0) {
    if (role == 0) {
      a.stats[0] = (float)early_stop_epoch;
      a.stats[1] = (float)steps_done;
      a.stats[2] = st_ent; a.stats[3] = st_pg; a.stats[6] = st_cf;
      a.stats[8] = last_loss;
      a.stats[11] = (float)status;
      a.adam_t[0] = t0 + steps_done;
    } else if (role == 1) {
      a.stats[4] = st_vl; a.stats[9] = last_loss;
    } else {
      a.stats[5] = st_vl; a.stats[10] = last_loss;
    }
  }
}

// flat env-major indices of the permutations (buffers.py:53-65: i -> env = i / T, t = i % T) -> [T,N] storage offsets, once per
// launch instead of once per gathered row inside the persistent kernel
__global__ void ppo_perm_offsets_kernel(const int* __restrict__ perms, long long n, int T, int N, int* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int idx = perms[i];
  const int env = idx / T, t = idx - env * T;
  out[i] = t * N + env;
}

__global__ void ppo_plan_kernel(const int* adam_t, int n_steps, int n_mb, int n_total, int B, double lr, double b1, double b2,
                                PlanStep* steps, PlanChunk* chunks, int two_per_step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_steps + 2) return;
  if (i >= n_steps) { steps[i] = PlanStep{0.f, 0.f, 0, 0}; return; }
  const int e = i / n_mb, mb = i % n_mb, p = mb * B;
  const int nb = n_total - p < B ? n_total - p : B;
  const int cpm = (B + RB - 1) / RB;
  const int nb_last = n_total - (n_mb - 1) * B;
  const int cpe = (n_mb - 1) * cpm + (nb_last + RB - 1) / RB;
  const int g0 = e * cpe + mb * cpm, nch = (nb + RB - 1) / RB;
  const double t = (double)(adam_t[0] + i + 1);
  steps[i] = PlanStep{(float)(lr / (1.0 - pow(b1, t))), (float)(1.0 / sqrt(1.0 - pow(b2, t))),
                      nb | ((mb == 0) << NB_FIRST) | ((mb == n_mb - 1) << NB_LAST) | (e << NB_EPOCH), e * n_total + p};
  if (chunks == nullptr) return;      // (the generic-shape path reads the step table only)
  if (two_per_step) {      // two workgroups per network: chunk c of step i at 2 i + c, an absent second chunk as 0 rows
    for (int c = 0; c < 2; ++c) chunks[2 * i + c] = c < nch ? PlanChunk{e * n_total + p + RB * c, nb - RB * c < RB ? nb - RB * c : RB} : PlanChunk{0, 0};
    if (i == n_steps - 1)
      for (int c = 0; c < 10; ++c) chunks[2 * n_steps + c] = PlanChunk{0, 0};
    return;
  }
  for (int c = 0; c < nch; ++c) chunks[g0 + c] = PlanChunk{e * n_total + p + RB * c, nb - RB * c < RB ? nb - RB * c : RB};
  if (i == n_steps - 1)
    for (int c = 0; c < 5; ++c) chunks[g0 + nch + c] = PlanChunk{0, 0};
}

}  // namespace icrl

using namespace icrl;

template <int NT1, bool DISC>
static int launch_train(const TrainArgs& a, hipStream_t s) {
  static_assert(Smem<NT1>::TOTAL * sizeof(float) <= 160 * 1024, "LDS budget");
  const size_t bytes = (size_t)Smem<NT1>::TOTAL * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)ppo_train_kernel<NT1, DISC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  TrainArgs arg = a;
  return (int)launch_coresident(ppo_train_kernel<NT1, DISC>, dim3(3), dim3(TH), bytes, s, arg);
}

// argument checks + the pre-kernels (granule / statistics reset, permutation offsets, schedule tables) of ONE run; fills `a` and
// reports which persistent kernel runs it: 0 wave pairs, 1 row-owning waves, 2 row-owning waves with two workgroups per network,
// 3 column-split tiles, 6 / 7 four workgroups per network at obs 65..128 (single-run launches; 7: minibatches of 65..128 rows in one pass), 4 wave quads with two workgroups per network (obs <= 32, launches of up to HALVES_MAX_RUNS runs: beyond that the
// compute units are what runs out and a run keeps one per network).  < 0: refused (return value of fail()) or a HIP error, in *err.
// FOUR workgroups per network where two would run (round 6: 6.46 against 6.74 us per optimiser step at HC shapes); ICRL_QUARTERS=0 keeps two (A/B)
static bool quarters_default() {
  static const int v = [] { const char* e = getenv("ICRL_QUARTERS"); return e != nullptr ? (e[0] == '1' ? 1 : 0) : 1; }();
  return v != 0;
}

static int prepare_train(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, const icrl_buffer_t* buf,
                         const int32_t* perms, const float* nu, const icrl_ppo_hyper_t* hp, float* stats, void* sync_ws,
                         hipStream_t s, TrainArgs& a, int* err, int n_runs) {
  auto bad = [&](int e) { *err = e; return -1; };
  if (pol->arch != nullptr || pol->h1 != HD || pol->h2 != HD)
    return bad(fail("icrl_ppo_lag_train: hidden widths (%d, %d)%s; the persistent update kernels are built for %d x %d (the reference's default net_arch)", pol->h1, pol->h2,
                    pol->arch != nullptr ? " with an `arch` descriptor" : "", HD, HD));
  if (pol->obs_dim < 1 || pol->obs_dim > 128 || pol->act_dim < 1 || pol->act_dim > 16)
    return bad(fail("icrl_ppo_lag_train: obs_dim %d (1..128) / act_dim %d (1..16)", pol->obs_dim, pol->act_dim));
  if (hp->batch_size < 2 || hp->batch_size > MAXB || hp->n_epochs < 1)
    return bad(fail("icrl_ppo_lag_train: batch_size %d (2..%d: one minibatch = at most four 64-row chunks of one workgroup), n_epochs %d (>= 1)", hp->batch_size, MAXB, hp->n_epochs));
  if (hp->n_epochs >= (1 << 19)) return bad(fail("icrl_ppo_lag_train: n_epochs %d, limit 2^19", hp->n_epochs));
  if (hp->batch_size > 128 && (hp->_pad & 2))
    return bad(fail("icrl_ppo_lag_train: batch_size %d: the column-split tiles kernel (hp->_pad & 2) stops at 128 rows", hp->batch_size));
  if (buf->obs_dim != pol->obs_dim || buf->T < 1)
    return bad(fail("icrl_ppo_lag_train: buffer obs_dim %d vs policy %d, T = %d", buf->obs_dim, pol->obs_dim, buf->T));
  if ((long long)buf->T * buf->N >= (1ll << 31) || (long long)buf->T * buf->N * (pol->obs_dim > 16 ? pol->obs_dim : 16) >= (1ll << 30))
    return bad(fail("icrl_ppo_lag_train: %d x %d transitions x obs_dim %d overflow the kernel's 32-bit element offsets (limit 2^30 floats per plane)", buf->T, buf->N, pol->obs_dim));
  if (pol->discrete && buf->act_store != 1) return bad(fail("icrl_ppo_lag_train: discrete policy needs act_store = 1 (action index), got %d", buf->act_store));
  a.L = make_pol_layout(pol->obs_dim, pol->act_dim, pol->h1, pol->h2, pol->discrete);
  a.params = pol->params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.adam_t = adam_step;
  a.buf = *buf; a.perms = perms; a.nu = nu; a.hp = *hp; a.stats = stats; a.xch = (u64*)sync_ws;
  a.t_magic = buf->T == 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned long long)buf->T);
  a.plan_steps = nullptr; a.plan_chunks = nullptr; a.n_steps = 0; a.gx = nullptr;
  hipError_t e = hipMemsetAsync(sync_ws, 0, 512, s);      // granule slots: 2 step parities x (3 roles x 8 waves, padded to 32) x 8 B
  if (e != hipSuccess) return bad((int)e);
  e = hipMemsetAsync(stats, 0, (32 + hp->n_epochs) * sizeof(float), s);
  if (e != hipSuccess) return bad((int)e);
  const int nt1 = (pol->obs_dim + 15) / 16;
  // default: wave pairs (two waves per SIMD, 6 barriers per step); hp._pad & 4: row-owning waves (one wave per SIMD, 3 barriers
  // per step; 5 when obs > 64); hp._pad & 2: the column-split tiles kernel
  if (nt1 > 8 || (hp->_pad & 2)) {
    if (hp->batch_size > 128) return bad(fail("icrl_ppo_lag_train: batch_size %d with obs_dim %d: the column-split tiles kernel stops at 128 rows", hp->batch_size, pol->obs_dim));
    return 3;
  }
  const int n_total = buf->T * buf->N;
  const int n_mb = (n_total + hp->batch_size - 1) / hp->batch_size;
  const long long n_steps = (long long)hp->n_epochs * n_mb;
  if (n_steps >= (1ll << 21)) return bad(fail("icrl_ppo_lag_train: %lld optimiser steps per call, limit 2^21", n_steps));      // epoch index lives in the upper bits of nb_flags
  PlanStep* steps = reinterpret_cast<PlanStep*>((char*)sync_ws + 512);
  PlanChunk* chunks = reinterpret_cast<PlanChunk*>(steps + n_steps + 2);
  a.plan_steps = steps; a.plan_chunks = chunks; a.n_steps = (int)n_steps;
  {
    // after the plan tables: 16 B per step (+2 entries) and 8 B per chunk (<= 4 chunks per step at batch_size 256, +10 entries):
    // 512 + 16 (n + 2) + 8 (4 n + 10) <= 768 + 48 n  (ICRL_PPO_PLAN_BYTES)
    int* offs = reinterpret_cast<int*>((char*)sync_ws + ICRL_PPO_PLAN_BYTES(n_steps));
    const long long n = (long long)hp->n_epochs * n_total;
    hipLaunchKernelGGL(ppo_perm_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perms, n, buf->T, buf->N, offs);
    a.perms = offs;
  }
  // obs <= 64: wave pairs; wider (AntWall 113: dz1^T must share h2^T's LDS, two chunks per minibatch) the row-owning waves,
  // with TWO workgroups per network when a minibatch has two chunks (each computes one, partial gradients exchanged;
  // hp._pad & 8 keeps one workgroup per network)
  // obs 65..128, single-run launches: FOUR workgroups per network, wave quads + the row-owning kernel's parameter ownership
  // (ppo_train_quarters.hip; round 6); hp._pad & 4 keeps the row-owning kernel, hp._pad & 32 / ICRL_QUARTERS=0 its two-workgroup form
  const bool quarters_wide = nt1 > 4 && n_runs == 1 && !(hp->_pad & (4 | 32)) && quarters_default();
  // ... minibatches of 65..128 rows (the reference's AntWall batch size): both 64-row chunks in ONE pass, two row tiles per wave (ppo_train_quarters2.hip;
  // the chunk plan with two entries per step); ICRL_QUARTERS_PASSES=1 keeps the chunk-by-chunk form (A/B)
  static const bool one_pass = [] { const char* e = getenv("ICRL_QUARTERS_PASSES"); return e == nullptr || e[0] != '1'; }();
  const bool quarters_wide2 = quarters_wide && one_pass && hp->batch_size > RB && hp->batch_size <= 2 * RB;
  const bool rows = !quarters_wide && ((hp->_pad & 4) || nt1 > 4);
  const bool split = rows && hp->batch_size > RB && hp->batch_size <= 2 * RB && !(hp->_pad & 8);     // (three or four chunks: one workgroup walks them)
  // hp._pad & 16: the wave-pair kernel (one workgroup per network) where the wave-quad kernel would run
  // (5: four workgroups per network — round 6; hp._pad & 32 keeps two)
  const bool halves = !rows && n_runs <= HALVES_MAX_RUNS && nt1 <= 2 && !(hp->_pad & 16);
  const bool quarters = halves && n_runs <= QUARTERS_MAX_RUNS && !(hp->_pad & 32) && quarters_default();
  hipLaunchKernelGGL(ppo_plan_kernel, dim3((unsigned)((n_steps + 2 + 255) / 256)), dim3(256), 0, s, adam_step, (int)n_steps, n_mb,
                     n_total, hp->batch_size, (double)hp->lr, (double)hp->adam_beta1, (double)hp->adam_beta2, steps, chunks, (int)(split || quarters_wide2));
  if (split || halves || quarters_wide) {
    const size_t off = (ICRL_PPO_PLAN_BYTES(n_steps) + 4 * (size_t)hp->n_epochs * n_total + 255) / 256 * 256;      // behind the permutation offsets
    a.gx = reinterpret_cast<u64*>((char*)sync_ws + off);
    e = hipMemsetAsync(a.gx, 0, ICRL_PPO_SPLIT_BYTES, s);
    if (e != hipSuccess) return bad((int)e);
  }
  if (quarters_wide) return quarters_wide2 ? 7 : 6;
  return rows ? (split ? 2 : 1) : (halves ? (quarters ? 5 : 4) : 0);
}

// shapes the persistent kernels refuse (hidden widths above 64, architectures given by icrl_policy_t.arch, minibatches above 256 rows):
// the generic-shape path of generic.hip, three plain launches per optimiser step.  Its scratch lies behind the regular workspace:
// sync_ws must then hold ICRL_PPO_SYNC_BYTES(...) + ICRL_PPO_GENERIC_BYTES(batch_size, row_floats, n_params) bytes.
static int train_generic(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, const icrl_buffer_t* buf,
                         const int32_t* perms, const float* nu, const icrl_ppo_hyper_t* hp, float* stats, void* sync_ws, hipStream_t s) {
  if (hp->batch_size < 2 || hp->n_epochs < 1) return fail("icrl_ppo_lag_train: batch_size %d (>= 2), n_epochs %d (>= 1)", hp->batch_size, hp->n_epochs);
  if (pol->arch != nullptr) { if (int e = policy_generic_check(pol, "icrl_ppo_lag_train")) return e; }
  else if (pol->h1 != pol->h2 || pol->h1 % 64 != 0 || pol->h1 > 256 || pol->obs_dim < 1 || pol->obs_dim > 1024 || pol->act_dim < 1 || pol->act_dim > 16)
    return fail("icrl_ppo_lag_train: hidden widths (%d, %d), obs_dim %d, act_dim %d: the persistent kernels are built for %d x %d (narrower layers stored "
                "zero-padded), obs <= 128, act <= 16; the generic-shape path for a common padded width that is a multiple of 64 up to 256, obs <= 1024",
                pol->h1, pol->h2, pol->obs_dim, pol->act_dim, HD, HD);
  if (buf->obs_dim != pol->obs_dim || buf->T < 1) return fail("icrl_ppo_lag_train: buffer obs_dim %d vs policy %d, T = %d", buf->obs_dim, pol->obs_dim, buf->T);
  if ((long long)buf->T * buf->N * (pol->obs_dim > 16 ? pol->obs_dim : 16) >= (1ll << 31))
    return fail("icrl_ppo_lag_train: %d x %d transitions x obs_dim %d overflow 32-bit element offsets", buf->T, buf->N, pol->obs_dim);
  if (pol->discrete && buf->act_store != 1) return fail("icrl_ppo_lag_train: discrete policy needs act_store = 1 (action index), got %d", buf->act_store);
  const int n_total = buf->T * buf->N;
  const int n_mb = (n_total + hp->batch_size - 1) / hp->batch_size;
  const long long n_steps = (long long)hp->n_epochs * n_mb;
  hipError_t e = hipMemsetAsync(stats, 0, (32 + hp->n_epochs) * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  int* offs = reinterpret_cast<int*>((char*)sync_ws + ICRL_PPO_PLAN_BYTES(n_steps));
  const long long n = (long long)hp->n_epochs * n_total;
  hipLaunchKernelGGL(ppo_perm_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perms, n, buf->T, buf->N, offs);
  char* scratch = (char*)sync_ws + ICRL_PPO_SYNC_BYTES(hp->n_epochs, n_mb, n_total);
  const int err = launch_train_generic(pol, exp_avg, exp_avg_sq, adam_step, buf, offs, nu, hp, stats, scratch, sync_ws, s);
  // the generic kernels keep THEIR transposed image in params_t; a policy the fast forward kernels serve (this call came here for its
  // batch size) gets the image those kernels read back
  if (err == 0 && !policy_is_wide(pol)) return icrl_policy_prepare(pol, (void*)s);
  return err;
}

extern "C" int icrl_ppo_generic_row_floats(const icrl_policy_t* pol) { return generic_row_floats(pol); }

extern "C" int icrl_ppo_lag_train(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step,
                                  const icrl_buffer_t* buf, const int32_t* perms, const float* nu,
                                  const icrl_ppo_hyper_t* hp, float* stats, void* sync_ws, void* stream) {
  TrainArgs a;
  hipStream_t s = (hipStream_t)stream;
  if (policy_is_wide(pol) || hp->batch_size > MAXB)
    return train_generic(pol, exp_avg, exp_avg_sq, adam_step, buf, perms, nu, hp, stats, sync_ws, s);
  int err = 0;
  const int kind = prepare_train(pol, exp_avg, exp_avg_sq, adam_step, buf, perms, nu, hp, stats, sync_ws, s, a, &err, 1);
  if (kind < 0) return err;
  const int nt1 = (pol->obs_dim + 15) / 16;
  if (kind == 4 || kind == 5) return launch_train_halves(a, pol->discrete != 0, kind == 5 ? 4 : 2, s);
  if (kind == 6) return launch_train_quarters_wide(a, pol->discrete != 0, s);
  if (kind == 7) return launch_train_quarters_wide2(a, pol->discrete != 0, s);
  if (kind == 0) return launch_train_pairs(a, nt1, pol->discrete != 0, s);
  if (kind <= 2) return launch_train_rows(a, nt1, pol->discrete != 0, kind == 2, s);
  if (pol->discrete) {
    if (nt1 <= 2) return launch_train<2, true>(a, s);
    if (nt1 <= 4) return launch_train<4, true>(a, s);
    return launch_train<8, true>(a, s);
  }
  if (nt1 <= 2) return launch_train<2, false>(a, s);
  if (nt1 <= 4) return launch_train<4, false>(a, s);
  return launch_train<8, false>(a, s);
}

extern "C" int icrl_ppo_lag_train_batch(int n_runs, const icrl_ppo_train_job_t* jobs, void* args_ws, long long args_ws_bytes,
                                        void* stream) {
  if (n_runs < 1 || n_runs > 65535) return fail("icrl_ppo_lag_train_batch: n_runs = %d (1..65535)", n_runs);
  if (args_ws == nullptr || args_ws_bytes < (long long)n_runs * (long long)sizeof(TrainArgs))
    return fail("icrl_ppo_lag_train_batch: args_ws holds %lld B, %d runs need %lld (ICRL_BATCH_ARGS_BYTES each)", args_ws_bytes, n_runs,
                (long long)n_runs * (long long)sizeof(TrainArgs));
  static_assert(sizeof(TrainArgs) <= ICRL_BATCH_ARGS_BYTES, "ICRL_BATCH_ARGS_BYTES");
  hipStream_t s = (hipStream_t)stream;
  TrainArgs* d_args = (TrainArgs*)args_ws;
  int kind0 = -1;
  const icrl_ppo_train_job_t& j0 = jobs[0];
  for (int r = 0; r < n_runs; ++r) {
    const icrl_ppo_train_job_t& j = jobs[r];
    if (j.pol->obs_dim != j0.pol->obs_dim || j.pol->act_dim != j0.pol->act_dim || j.pol->discrete != j0.pol->discrete ||
        j.hp->batch_size != j0.hp->batch_size || j.hp->n_epochs != j0.hp->n_epochs || j.hp->_pad != j0.hp->_pad ||
        j.buf->T != j0.buf->T || j.buf->N != j0.buf->N || j.buf->act_store != j0.buf->act_store)
      return fail("icrl_ppo_lag_train_batch: run %d differs from run 0 in a shape (obs / act / discrete / batch_size / n_epochs / T / N): the runs of a batch share one grid", r);
    TrainArgs a;
    int err = 0;
    const int kind = prepare_train(j.pol, j.exp_avg, j.exp_avg_sq, j.adam_step, j.buf, j.perms, j.nu, j.hp, j.stats, j.sync_ws, s, a, &err, n_runs);
    if (kind < 0) return err;
    if (kind == 3) return fail("icrl_ppo_lag_train_batch: the column-split tiles kernel (hp->_pad & 2) has no batched form");
    if (r == 0) kind0 = kind;
    if (kind == 6) return launch_train_quarters_wide(a, j.pol->discrete != 0, s);      // (a one-run batch at obs 65..128: the single-run launch)
    if (kind == 7) return launch_train_quarters_wide2(a, j.pol->discrete != 0, s);
    const int e = put_args(a, d_args + r, s);
    if (e != 0) return e;
  }
  const int nt1 = (j0.pol->obs_dim + 15) / 16;
  if (kind0 == 4 || kind0 == 5) return launch_train_halves_batch(d_args, n_runs, j0.pol->obs_dim, j0.pol->discrete != 0, kind0 == 5 ? 4 : 2, (j0.hp->_pad & 1) != 0, s);
  if (kind0 == 0) return launch_train_pairs_batch(d_args, n_runs, j0.pol->obs_dim, nt1, j0.pol->discrete != 0, (j0.hp->_pad & 1) != 0, s);
  return launch_train_rows_batch(d_args, n_runs, nt1, j0.pol->discrete != 0, kind0 == 2, s);
}
