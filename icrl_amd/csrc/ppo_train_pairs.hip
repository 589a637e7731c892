// PPO-Lagrangian update, wave-PAIR variant (obs_dim <= 128) — gfx950.
//
// ref: stable_baselines3/ppo_lag/ppo_lag.py:196-299, common/buffers.py:594-627, common/policies.py:752-767,
//      common/distributions.py:143-171,274-288, torch.optim.Adam, clip_grad_norm_  (same contract as ppo_train_rows.hip).
//
// Same launch shape (3 persistent workgroups = pi | vf | cvf), exchange protocol, LDS images and transposed-GEMM trick as
// ppo_train_rows.hip, but EIGHT waves per workgroup = two per SIMD.  Why: the fp32 MFMA (v_mfma_f32_16x16x4_f32) executes on
// the SIMD's fp32 FMA lanes, so a wave's own VALU work never hides under its MFMAs, and a single wave per SIMD issues a VALU
// instruction only every 4 cycles (the SIMD takes one every 2): with one wave per SIMD a step is
// MFMA time + VALU time + every LDS / transcendental latency, fully exposed.  Here the two waves of a SIMD split the work of
// one 16-row tile so that
//   * the MFMA pipe sees the same ~300 instructions per step (no redundant GEMM work),
//   * everything else (tanh, loss tails, stores, Adam, operand fetches) is halved per wave and the two halves issue in
//     alternate cycles, and one wave's LDS / MFMA-result latency is covered by the other's instructions,
//   * each wave needs half the registers (<= 256, the limit at two waves per SIMD): no accumulation-register shuffling.
//
// Wave w = (rt = w % 4, fh = w / 4); waves (rt, 0) and (rt, 1) sit on the same SIMD and own minibatch rows 16 rt .. 16 rt + 15.
//   forward / activation backward: wave (rt, fh) computes the feature tiles t in {2 fh, 2 fh + 1} of every layer for its rows;
//     the partner's half of h1 / dz2 (the K dimension of the next GEMM) is read back from the [feature][row] images both
//     waves store anyway for the weight-gradient GEMMs; the 16-output head is split over K instead (8 MFMAs each, partial
//     tiles exchanged through LDS) and both waves then evaluate the loss tail on the full head output;
//   weight gradients (K = all 64 rows): wave (jt = rt, kh = fh) forms rows 16 jt.. of dW2 for the column tiles {2 kh, 2 kh + 1},
//     of dW1 for its half of the observation tiles, and (kh = 0) columns 16 jt.. of dWh; it owns exactly those elements'
//     Adam state.
//   3 pair hand-offs (h1 | head partials | dz2) and 3 workgroup barriers (dz1 | norm + staging | weights) per optimiser step.
//
// Built with -ffp-contract=off; FMA is used only where written (fmaf / MFMA).
#include "ppo_common.h"

// A/B switches (tools/build_variant.sh): the defaults are what ships
#ifndef ICRL_EARLY_POLL
#define ICRL_EARLY_POLL 0
#endif
#ifndef ICRL_STATIC_LDS
#define ICRL_STATIC_LDS 1
#endif
#ifndef ICRL_BOOK_WAVE
#define ICRL_BOOK_WAVE 7
#endif
#ifndef ICRL_W1_TAIL
#define ICRL_W1_TAIL 0
#endif
// weight-gradient GEMMs: ALL operand tiles of a GEMM fetched before its first MFMA (a scheduling barrier; left alone the compiler fetches
// one K step at a time and waits for the LDS in front of every group of four dependent MFMAs).  Measured: 8.53 vs 8.53 us — the other
// wave of the SIMD already covers those waits; off.
#ifndef ICRL_OPERANDS_FIRST
#define ICRL_OPERANDS_FIRST 0
#endif
#if ICRL_OPERANDS_FIRST
#define OPERANDS_FIRST() __builtin_amdgcn_sched_barrier(0)
#else
#define OPERANDS_FIRST()
#endif
#ifndef ICRL_EARLY_COMMIT
#define ICRL_EARLY_COMMIT 0
#endif
#ifndef ICRL_LOW_GATHER
#define ICRL_LOW_GATHER 0
#endif
#ifndef ICRL_HIGH_PRIO
#define ICRL_HIGH_PRIO 0
#endif
#ifndef ICRL_LOW_PRIO
#define ICRL_LOW_PRIO 0
#endif
#ifndef ICRL_W1_TAIL_MOVE_HEAD
#define ICRL_W1_TAIL_MOVE_HEAD 0
#endif
#ifndef ICRL_L1_TAILQ
#define ICRL_L1_TAILQ 1
#endif
// head outputs along the lane groups (see `out_of`): the dH2 GEMM then needs ceil(n_out / 4) of its 4 MFMAs per tile
#ifndef ICRL_HEAD_PERM
#define ICRL_HEAD_PERM 1
#endif
#ifndef ICRL_STATS_WAVE0
#define ICRL_STATS_WAVE0 1
#endif
#ifndef ICRL_L1_AHEAD
#define ICRL_L1_AHEAD 0
#endif

// Timing diagnostics (tools/diag_train.sh; WRONG RESULTS, never the shipped build): -DICRL_DIAG=<bits> removes one component of the
// step so that its marginal cost on the critical path can be read off the step time:
//   1 barrier S5 | 2 barrier S6 | 4 barrier S7 | 8 pair hand-offs | 16 tanh | 32 Adam arithmetic | 64 loss tail | 128 row prefetch + staging
//   256 gradient norm (partial sums, publish, poll) | 512 advantage statistics | 1024 book-keeping lane
#ifndef ICRL_DIAG
#define ICRL_DIAG 0
#endif

namespace icrl {

#if ICRL_DIAG & 16
#define fast_tanh(x) ((x) * 0.5f)
#endif

constexpr int TH8 = 512;   // 8 waves, two per SIMD (<= 256 registers each)
constexpr int ST = 72;     // row stride of the [feature][row] matrices (64 rows + 8: conflict-free ds_read_b128)
constexpr int SA = 24;     // row stride of the per-row action block and of the transposed head weights

template <int NT1>
struct SmemP {  // offsets in floats (multiples of 4); same images as SmemR in ppo_train_rows.hip
  static constexpr int O16 = 16 * NT1, SX = O16 + 8;
  static constexpr bool XDB = NT1 <= 2;        // second X buffer (next minibatch staged while dW1 still reads this one)
  static constexpr bool W2TC = NT1 <= 2;       // transposed copy of W2 (ds_read_b128 operand fetch in the backward); LDS budget
  static constexpr bool WHTC = NT1 <= 4;       // transposed copy of the head weights
  static constexpr bool DZ1A = NT1 > 4;        // dz1^T shares h2^T's storage (written after dWh has read h2^T): LDS budget
  static constexpr int W1 = 0;                 // [64][SX]
  static constexpr int W2 = W1 + HD * SX;      // [64][SH]
  static constexpr int W2T = W2 + HD * SH;     // [64][SH]  W2T[k][j] = W2[j][k]
  static constexpr int WH = W2T + (W2TC ? HD * SH : 0);   // [16][SH]
  static constexpr int WHT = WH + 16 * SH;     // [64][SA]  WHT[j][o] = WH[o][j]
  static constexpr int B1 = WHT + (WHTC ? HD * SA : 0);
  static constexpr int B2 = B1 + HD;
  static constexpr int BH = B2 + HD;
  static constexpr int LS = BH + 16;
  static constexpr int GAU = LS + 16;          // [3][16] per-action 1/var, 0.5/var, log(sd) + log(sqrt(2 pi))
  static constexpr int XT0 = GAU + 48;         // [16 NT1][ST] x^T of the chunk: XT[k][row]
  static constexpr int XT1 = XDB ? XT0 + O16 * ST : XT0;
  static constexpr int H1T = XT1 + O16 * ST;   // [64][ST] h1^T
  static constexpr int H2T = H1T + HD * ST;
  static constexpr int DZ1T = DZ1A ? H2T : H2T + HD * ST;
  static constexpr int DZ2T = H2T + (DZ1A ? 1 : 2) * HD * ST;
  static constexpr int DOT = DZ2T + HD * ST;   // [16][ST] d loss / d head output, transposed
  static constexpr int ACT = DOT + 16 * ST;    // [64][SA] actions of the chunk's rows
  static constexpr int OLP = ACT + RB * SA;    // [64] old log-prob | old value
  static constexpr int ADR = OLP + RB;         // [64] raw reward advantage | return
  static constexpr int ADC = ADR + RB;         // [64] raw cost advantage
  static constexpr int PST = ADC + RB;         // [4][8] per-row-tile loss statistics
  static constexpr int PLS = PST + 32;         // [4][16] per-row-tile d log_std partial sums
  static constexpr int MISC = PLS + 64;        // [64] granule values, flags, advantage-statistics partials
  static constexpr int TOTAL = MISC + 64;
  // head partial tiles of a pair (2 x [64 lanes] x f32x4) live in the pair's own 16 columns of the dz1^T image (rows 0..31):
  // that image is written by the same pair only after both waves have read the partials (behind the pair's dz2 hand-off).
  static constexpr int HPX = DZ1T;
};

// -DICRL_FINE_PROF (tools only, never the shipped build): 20 finer phase timers of the policy workgroup instead of the 7 coarse ones
#ifdef ICRL_FINE_PROF
#undef STAMP
#define STAMP(slot)
#define FSTAMP(k)                                                          \
  if (prof) {                                                              \
    __builtin_amdgcn_sched_barrier(0);                                     \
    const unsigned long long now_ = stamp();                               \
    __builtin_amdgcn_sched_barrier(0);                                     \
    fph[k] += now_ - t_last;                                               \
    t_last = now_;                                                         \
  }
#else
#define FSTAMP(k)
#endif

// re-read a rarely used argument from the launch's argument block instead of keeping it in a register for the whole loop
#define KARGS() ([&]() { const TrainArgs* k_ = ka; asm volatile("" : "+s"(k_)); return k_; }())

// The whole update of ONE run: `a` are its arguments, `ka` points at the same block in memory (the kernel-argument segment of a
// single-run launch, element blockIdx.y of the device-side argument array of a batched launch).
// OBS > 0: the observation width is known at compile time (the widths of the BASELINE configs get their own instantiation): layer-1
// MFMAs whose four k values are all padding (k = 16 js + 4 q + e >= obs for every q, i.e. 16 js + e >= obs) are not issued — at
// obs 18 six of eight per tile remain, at obs 1 (LapGridWorld) one.  OBS == 0: every k step runs against the zero pad weights.
// PROF: the diagnostic phase timers (hp._pad & 1) as a compile-time variant (ppo_train_halves.hip: as a run-time flag they cost every launch ~2.5 %)
template <int NT1, bool DISC, int OBS = 0, bool PROF = false>
__device__ __forceinline__ void ppo_train_pairs_body(const TrainArgs& a, const TrainArgs* const ka, const int role_arg) {
  using S = SmemP<NT1>;
  constexpr int SX = S::SX;
  constexpr int NW1 = NT1 / 2;          // observation column tiles of dW1 per wave
  constexpr bool EARLY_COMMIT = ICRL_EARLY_COMMIT && S::XDB;      // needs the second X^T buffer
  constexpr int XR = (S::O16 + 7) / 8; (void)XR;  // floats of an X row each of the 8 threads of a row stages
  static_assert(NT1 % 2 == 0, "the two waves of a pair split the observation tiles");
  static_assert(!S::DZ1A, "wide observations (dz1^T sharing h2^T's storage) stay on the row-owning kernel");
#if ICRL_STATIC_LDS
  // a STATIC array: its address is the compile-time constant 0, so every image offset folds into an instruction's immediate or one
  // literal move.  With `extern __shared__` the base is a symbol the optimiser cannot fold: it hoisted ~40 "base + offset" sums out of
  // the step loop into scalar registers, spilled them to VGPR lanes and read them back (107 v_readlane + s_nop + v_mov per step)
  __shared__ __attribute__((aligned(16))) float sm[S::TOTAL];
#else
  extern __shared__ __attribute__((aligned(16))) float sm[];
#endif
  const int role = role_arg;  // 0 policy, 1 reward critic, 2 cost critic
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Which two waves form a pair, and how they hand data to each other, is a speed choice only (nothing depends on the
  // placement).  Measured on MI355X, HC shapes, us per optimiser step (the row-owning kernel: 10.5):
  //   partners w, w + 4 (a workgroup's waves go to the SIMDs round-robin: the partners SHARE a SIMD), LDS flags   9.6
  //   partners w, w + 4, workgroup barriers instead of flags                                                       9.9
  //   partners w, w ^ 1 (different SIMDs), workgroup barriers 10.6; flags 10.4
#ifndef ICRL_OWN_LDS
#define ICRL_OWN_LDS 1
#endif
#ifndef ICRL_PAIR_ADJ
#define ICRL_PAIR_ADJ 0
#endif
#ifndef ICRL_PAIR_FLAGS
#define ICRL_PAIR_FLAGS 1
#endif
#if ICRL_PAIR_ADJ
  const int rt = w >> 1, fh = w & 1;       // row tile, feature half (= jt, kh in the weight-gradient phase)

  const int partner = w ^ 1;
#else
  const int rt = w & 3, fh = w >> 2;

  const int partner = w ^ 4;
#endif
  const int r = lane & 15, q = lane >> 4;
  const int O = a.L.O, A = a.L.A;
  const int n_out = role == 0 ? A : 1;
  // Where a head output sits in the 16-wide M / K index of the head GEMMs ("position").  An MFMA's C layout gives lane (r, q) the
  // positions 4 q + i (i = 0..3), and as a K index MFMA e covers the positions 4 q + e of its four lane groups.  With output o at
  // position p = 4 (o % 4) + o / 4 (a 4 x 4 transpose, its own inverse) lane (r, q) holds the outputs o = 4 i + q, and MFMA e of
  // dH2 = Wh^T . dOut covers the outputs 4 e .. 4 e + 3: the MFMAs with 4 e >= n_out multiply zeros and are not issued (6 actions:
  // 2 of 4; a critic: 1 of 4).  Every head image (WH, WHT, BH, LS, GAU, ACT, DOT, PLS) is indexed by position; only the places
  // that meet a real output index (parameter load / write-back, masks against n_out, the staged action columns) go through
  // out_of / pos_of.
  auto out_of = [&](int i) { return ICRL_HEAD_PERM ? 4 * i + q : 4 * q + i; };      // output of this lane's C-layout element i
  auto pos_of = [](int o) { return ICRL_HEAD_PERM ? 4 * (o & 3) + (o >> 2) : o; };   // position of output o (and back)
  const int ro = pos_of(r);                                                          // the output at position r
  const int T = a.buf.T, N = a.buf.N;
  const float nu = as_global(a.nu)[0];
  const int n_steps = a.n_steps;
  const PlanStep* __restrict__ const plan_steps = as_global(a.plan_steps);
  const PlanChunk* __restrict__ const plan_chunks = as_global(a.plan_chunks);
  const int* __restrict__ const perms = as_global(a.perms);
  const float* const p_s0 = as_global(role == 0 ? a.buf.log_probs : (role == 1 ? a.buf.reward_values : a.buf.cost_values));
  const float* const p_s1 = as_global(role == 0 ? a.buf.reward_advantages : (role == 1 ? a.buf.reward_returns : a.buf.cost_returns));
  const float* const p_s2 = as_global(a.buf.cost_advantages);
  const float* const p_obs = as_global(a.buf.observations);
  const float* const p_act = as_global(a.buf.actions);
  const int AS = a.buf.act_store;

  // ---- Adam ownership of wave (jt = rt, kh = fh): element (row j = 16 jt + 4 q + i, column k = 16 c + r) of
  //   W2 for c in {2 kh, 2 kh + 1};  W1 for c in {NW1 kh .. NW1 kh + NW1 - 1};  and for kh == 0 only: head-weight element
  //   (o = 4 q + i, j = 16 jt + r), b1 / b2 entry 16 jt + r (replicated over q, lane q == 0 stores), wave 0: head bias r,
  //   wave 1: log_std r.  Master weights live in LDS (the operand images), the moments and the accumulating gradients in
  //   registers.
  f32x4 mW1[NW1], vW1[NW1], gW1r[NW1], mW2[2], vW2[2], gW2r[2], mWh, vWh, gWhr;
  const int jb = 16 * rt + r;
  float mb1 = 0.f, vb1 = 0.f, mb2 = 0.f, vb2 = 0.f, mex = 0.f, vex = 0.f, gb1r = 0.f, gb2r = 0.f, gex = 0.f;
  int ex_g = -1, ex_s = S::MISC + 63;
  const bool low = fh == 0;     // wave-uniform: owns b1 / b2 (and, without TAILW, the head column block and the extra entries)
  // TAILW (obs 17 / 18: the second observation tile holds one or two real columns) — MEASURED AND REJECTED, off by default.  That tile is a full
  // 16 x 16 x 64 GEMM of the high waves (16 MFMAs for <= 2 useful columns) and a full Adam block.  With TAILW the tail columns are per-row
  // vectors like the biases: d W1[j][16 + c] = sum_rows dz1^T[j][row] x^T[16 + c][row] as 16 FMAs per lane and column + the quad-row sum,
  // owned (q-replicated, lane q == 0 stores) by the high wave of row block rt: 16 MFMAs less on the SIMD's pipe per step.  8.48 us per
  // step against 8.42 — the ~45 VALU / LDS instructions cost the late (high) wave more than the 16 MFMAs did.  ICRL_W1_TAIL_MOVE_HEAD
  // additionally moves the head (dWh GEMM, its Adam block, head bias, log_std) to the high waves so that both waves of a pair carry 48
  // weight-gradient MFMAs and 4 Adam blocks (64 / 48 and 5 / 3 otherwise): 8.84 us — the high wave of a pair loses the issue arbitration
  // to the older low wave, so the uneven split IS the balanced one (per-wave timers: the high waves then reach the norm barrier 1.8 k
  // cycles after the low waves instead of 1 k before them).
  constexpr bool TAILW = ICRL_W1_TAIL && OBS > 0 && NT1 == 2 && OBS / 16 == 1 && (OBS % 16 == 1 || OBS % 16 == 2);
  constexpr int NTAIL = TAILW ? OBS % 16 : 0;
  const bool own_wh = (TAILW && ICRL_W1_TAIL_MOVE_HEAD) ? !low : low;          // head column block, head bias / log_std
  auto wave_of = [](int rt_, int fh_) { return ICRL_PAIR_ADJ ? 2 * rt_ + fh_ : rt_ + 4 * fh_; };
  const int W_BH = wave_of(0, (TAILW && ICRL_W1_TAIL_MOVE_HEAD) ? 1 : 0), W_LS = wave_of(1, (TAILW && ICRL_W1_TAIL_MOVE_HEAD) ? 1 : 0);       // the waves that own the head bias / log_std
  float mt[2] = {0.f, 0.f}, vt[2] = {0.f, 0.f}, gt[2] = {0.f, 0.f};                     // tail columns' moments / gradients (high waves)
  const bool own_w1_tile = !(TAILW && !low);       // this wave owns an observation tile of W1 (TAILW: the low waves' tile 0 only)
  auto w1_addr = [&](int cc, int i) { return S::W1 + (16 * rt + 4 * q + i) * SX + 16 * (NW1 * fh + cc) + r; };
  auto w2_addr = [&](int cc, int i) { return S::W2 + (16 * rt + 4 * q + i) * SH + 16 * (2 * fh + cc) + r; };
  auto store_w1 = [&](int cc, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[w1_addr(cc, i)] = v[i];
  };
  auto load_own_w1 = [&](int cc) -> f32x4 {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[w1_addr(cc, i)];
    return v;
  };
  auto store_w2 = [&](int cc, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[w2_addr(cc, i)] = v[i];
    if (S::W2TC) *reinterpret_cast<f32x4*>(sm + S::W2T + (16 * (2 * fh + cc) + r) * SH + 16 * rt + 4 * q) = v;
  };
  auto load_own_w2 = [&](int cc) -> f32x4 {
    if (S::W2TC) return lds128(sm + S::W2T + (16 * (2 * fh + cc) + r) * SH + 16 * rt + 4 * q);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[w2_addr(cc, i)];
    return v;
  };
  auto store_wh = [&](const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[S::WH + (4 * q + i) * SH + 16 * rt + r] = v[i];
    if (S::WHTC) *reinterpret_cast<f32x4*>(sm + S::WHT + (16 * rt + r) * SA + 4 * q) = v;
  };
  auto load_own_wh = [&]() -> f32x4 {
    if (S::WHTC) return lds128(sm + S::WHT + (16 * rt + r) * SA + 4 * q);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[S::WH + (4 * q + i) * SH + 16 * rt + r];
    return v;
  };
  for (int i = tid; i < S::TOTAL; i += TH8) sm[i] = 0.f;
  __syncthreads();
  {
    const PolLayout& L = a.L;
    const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
    const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
    const int gbh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);
#pragma unroll
    for (int cc = 0; cc < NW1; ++cc) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * rt + 4 * q + i, k = 16 * (NW1 * fh + cc) + r;
        const bool mine = own_w1_tile && k < O;
        pv[i] = mine ? a.params[gW1 + j * O + k] : 0.f;
        mW1[cc][i] = mine ? a.exp_avg[gW1 + j * O + k] : 0.f;
        vW1[cc][i] = mine ? a.exp_avg_sq[gW1 + j * O + k] : 0.f;
      }
      if (own_w1_tile) store_w1(cc, pv);
    }
    if (TAILW && !low) {
#pragma unroll
      for (int c = 0; c < NTAIL; ++c) {
        const int e = gW1 + jb * O + 16 + c;
        mt[c] = a.exp_avg[e]; vt[c] = a.exp_avg_sq[e];
        if (q == 0) sm[S::W1 + jb * SX + 16 + c] = a.params[e];
      }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * rt + 4 * q + i, k = 16 * (2 * fh + cc) + r;
        pv[i] = a.params[gW2 + j * HD + k];
        mW2[cc][i] = a.exp_avg[gW2 + j * HD + k];
        vW2[cc][i] = a.exp_avg_sq[gW2 + j * HD + k];
      }
      store_w2(cc, pv);
    }
    mWh = vWh = gWhr = f32x4{0.f, 0.f, 0.f, 0.f};
    if (own_wh) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = out_of(i), j = 16 * rt + r;
        pv[i] = o < n_out ? a.params[gWh + o * HD + j] : 0.f;
        mWh[i] = o < n_out ? a.exp_avg[gWh + o * HD + j] : 0.f;
        vWh[i] = o < n_out ? a.exp_avg_sq[gWh + o * HD + j] : 0.f;
      }
      store_wh(pv);
      if (w == W_BH && ro < n_out) { ex_g = gbh + ro; ex_s = S::BH + r; }      // lane r: the entry at POSITION r
      if (w == W_LS && !DISC && role == 0 && ro < A) { ex_g = L.log_std + ro; ex_s = S::LS + r; }
      if (ex_g >= 0) { mex = a.exp_avg[ex_g]; vex = a.exp_avg_sq[ex_g]; }
      if (q == 0) sm[ex_s] = ex_g >= 0 ? a.params[ex_g] : 0.f;        // lanes without an extra entry hit a scratch word
    }
    if (low) {
      mb1 = a.exp_avg[gb1 + jb]; vb1 = a.exp_avg_sq[gb1 + jb];
      mb2 = a.exp_avg[gb2 + jb]; vb2 = a.exp_avg_sq[gb2 + jb];
      if (q == 0) { sm[S::B1 + jb] = a.params[gb1 + jb]; sm[S::B2 + jb] = a.params[gb2 + jb]; }
    }
  }

  const int t0 = a.adam_t[0];
  const float w1 = (float)(1.0 - (double)a.hp.adam_beta1);
  const float w2 = (float)(1.0 - (double)a.hp.adam_beta2);
  const float clip = a.hp.clip_range;
  const float vclip = role == 1 ? a.hp.clip_range_reward_vf : a.hp.clip_range_cost_vf;
  const float vcoef = role == 1 ? a.hp.reward_vf_coef : a.hp.cost_vf_coef;
  const float ent_coef = a.hp.ent_coef;
  const float max_grad_norm = a.hp.max_grad_norm, adam_epsf = a.hp.adam_eps, adam_b2f = a.hp.adam_beta2;
  u64* const xch = as_global(a.xch);

  // ---------------------------------------------------------------------------------------------------------------
  // row stream (see ppo_train_rows.hip): `perms` holds storage offsets; rows of chunk g + 1 are prefetched into registers
  // while chunk g is processed.  The 16 rows of tile rt are staged by the 128 threads of the wave pair (rt, *): 8 per row.
  // ---------------------------------------------------------------------------------------------------------------
  // ICRL_LOW_GATHER (measured and rejected: 8.60 us per step against 8.49; off): the LOW wave of a pair fetches and stages all 16 rows of
  // the tile (4 threads per row) and the high wave none — in the forward the high wave is the late one of the pair (the low wave waits
  // ~1 k cycles for it behind the head partials) and the issue of the row fetches sits on its path; but the low wave then carries all of
  // the staging at the end of the step, where IT is the late one
  constexpr bool LOWG = ICRL_LOW_GATHER;
  constexpr int GP = LOWG ? 4 : 8;       // threads per row
  const int gb_row = 16 * rt + (lane >> 2), gpart = LOWG ? (lane & 3) : (lane & 3) + 4 * fh;
  const bool gather = !LOWG || fh == 0;  // wave-uniform
  // advantage statistics: row stid of the minibatch lives in waves SW0 .. SW0 + 3 (<= 256 rows).  SW0 = 1 (low waves 1, 2 at <= 128 rows):
  // measured 8.40 us per step against 8.45 with the statistics on the high waves 4, 5 — the younger wave of a pair already trails its
  // partner through every phase; extra work belongs on the leading one (wave 0 polls the granules, so the rows start at wave 1).
  constexpr int SW0 = ICRL_STATS_WAVE0;
  const int stid = tid - 64 * SW0;
  auto ld_step = [&](int i) -> int4 {
    asm volatile("" : "+v"(i));
    return *reinterpret_cast<const int4*>(plan_steps + i);
  };
  auto ld_chunk = [&](int g) -> int2 {
    asm volatile("" : "+v"(g));
    return *reinterpret_cast<const int2*>(plan_chunks + g);
  };
  // Every load of the two index streams is UNCONDITIONAL, at a clamped (always valid) position, and its value is not touched
  // before the step that consumes it: a load inside an exec-masked branch, or a select on the fresh value, makes the compiler
  // wait for ALL outstanding vector-memory operations right there (s_waitcnt vmcnt(0)) — i.e. for the row gathers issued a
  // few instructions earlier, a full trip to the Infinity Cache (~1.2 k cycles per step, measured).  Rows beyond the minibatch
  // simply re-read its first row; they are masked where it matters (`valid`, stats_partials).
  auto chunk_idx = [&](const int2& c) -> int { return perms[c.x + (gb_row < c.y ? gb_row : 0)]; };      // {perm_base, rows}
  auto stat_idx = [&](const int4& p) -> int {
    const int nbp = p.z & NB_MASK;
    return perms[p.w + ((stid >= 0 && stid < nbp) ? stid : 0)];
  };
  // Rows beyond the minibatch (idx < 0) and observation components beyond obs are fetched from a clamped, always valid address
  // and staged as they are: a row b >= nrows contributes nothing (its d loss / d output is forced to 0 by `valid`), and the
  // k >= obs columns of X only ever meet the zero pad columns of W1 (the dW1 step below keeps those at zero).  So the staging
  // path carries no masks and no branches; offsets are 32-bit element indices (the C ABI checks T * N * obs < 2^30).
  // Lean form: (i) with the observation width known at compile time only the pieces that hold real components are fetched and
  // staged (obs 18: 3 of 4 per thread; the X^T rows k >= obs keep the zeros of the start), (ii) the second action piece only when
  // there are more than 8 actions, (iii) ONE of the three per-row scalars per thread — thread gpart of a row fetches scalar gpart
  // (old log-prob / value | advantage / return | cost advantage) through its own pointer and stores it at its own address, threads
  // 3..7 re-read the third and park it in a scratch word.  9 -> 5 loads and 9 -> 5 stores per thread and step at HC shapes.
  constexpr int XRG = (S::O16 + GP - 1) / GP;
  constexpr int XRL = OBS > 0 ? (OBS + GP - 1) / GP : XRG;
  constexpr int NACT = 16 / GP;
  float px[XRG], pact[NACT], psc = 0.f;
#pragma unroll
  for (int i = 0; i < NACT; ++i) pact[i] = 0.f;
  const float* const p_sc = gpart == 0 ? p_s0 : (gpart == 1 ? p_s1 : p_s2);
  const int sc_dst = gpart == 0 ? S::OLP + gb_row : (gpart == 1 ? S::ADR + gb_row : (gpart == 2 ? S::ADC + gb_row : S::MISC + 62));
  auto issue_rows = [&](int idx) {
    if ((ICRL_DIAG & 128) || !gather) return;
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;
    const unsigned ob = off * (unsigned)O;
#pragma unroll
    for (int i = 0; i < XRL; ++i) { const int k = gpart + GP * i; px[i] = p_obs[ob + (unsigned)(k < O ? k : O - 1)]; }
    const unsigned ab = off * (unsigned)AS;      // (the critics fetch the action bytes too: a load is cheaper than a branch in the load stream)
#pragma unroll
    for (int i = 0; i < NACT; ++i)
      if (i == 0 || AS > GP * i) { const int k = gpart + GP * i; pact[i] = p_act[ab + (unsigned)(k < AS ? k : AS - 1)]; }
    psc = p_sc[off];
  };
  auto commit_rows = [&](int xbase) {
    if ((ICRL_DIAG & 128) || !gather) return;
#pragma unroll
    for (int i = 0; i < XRL; ++i) { const int k = gpart + GP * i; if (OBS > 0 ? k < OBS : k < S::O16) sm[xbase + k * ST + gb_row] = px[i]; }
    if (role == 0) {
#pragma unroll
      for (int i = 0; i < NACT; ++i)      // pad actions are 0; pieces beyond the action count are not written: those columns keep the zeros of the start
        if (i == 0 || AS > GP * i) { const int k = gpart + GP * i; sm[S::ACT + gb_row * SA + pos_of(k)] = k < AS ? pact[i] : 0.f; }
    }
    sm[sc_dst] = psc;
  };
  float sar = 0.f, sac = 0.f;
  auto issue_stats = [&](int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;
    sar = p_s1[off];
    sac = p_s2[off];
  };
  const bool big_mb = a.hp.batch_size > 128;     // minibatches of 129..256 rows: all four statistics waves hold rows
  auto stats_partials = [&](int nb) {
    if (ICRL_DIAG & 512) return;
    if (role != 0 || w < SW0 || w > SW0 + (big_mb ? 3 : 1)) return;
    const bool in = stid < nb;
    const float s_r = wave_sum_fast(in ? sar : 0.f), s_c = wave_sum_fast(in ? sac : 0.f), s_rr = wave_sum_fast(in ? sar * sar : 0.f);
    if (lane == 0) { sm[S::MISC + 3 * (w - SW0)] = s_r; sm[S::MISC + 3 * (w - SW0) + 1] = s_c; sm[S::MISC + 3 * (w - SW0) + 2] = s_rr; }
  };
  float mean_r = 0.f, istd_r = 1.f, mean_c = 0.f;
  auto read_stats = [&](int nb) {
    if (role != 0) return;
    // MISC[0..11]: (sum A_r, sum A_c, sum A_r^2) of the four statistics waves; the last two leave zeros for minibatches of <= 128 rows
    float s_r = sm[S::MISC + 0] + sm[S::MISC + 3];
    float s_c = sm[S::MISC + 1] + sm[S::MISC + 4];
    float s_rr = sm[S::MISC + 2] + sm[S::MISC + 5];
    if (big_mb) {
      s_r += sm[S::MISC + 6] + sm[S::MISC + 9];
      s_c += sm[S::MISC + 7] + sm[S::MISC + 10];
      s_rr += sm[S::MISC + 8] + sm[S::MISC + 11];
    }
    const float inv = __builtin_amdgcn_rcpf((float)nb);
    mean_r = s_r * inv;
    mean_c = s_c * inv;
    const float var = fmaxf(s_rr - s_r * mean_r, 0.f) * __builtin_amdgcn_rcpf((float)(nb - 1));
    istd_r = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) + 1e-8f);
  };
  auto refresh_gauss = [&]() {   // wave 1, lanes q == 0 own log_std r: derived constants of the Gaussian head
    if (!DISC && role == 0 && w == W_LS && q == 0) {
      const float wex = sm[S::LS + r];
      const float sd = __expf(wex);
      const float iv = __builtin_amdgcn_rcpf(sd * sd);
      sm[S::GAU + r] = ro < A ? iv : 0.f;
      sm[S::GAU + 16 + r] = ro < A ? 0.5f * iv : 0.f;
      sm[S::GAU + 32 + r] = ro < A ? wex + LOG_SQRT_2PI_F : 0.f;
      const float ent = row_sum(ro < A ? HALF_LOG_2PI_PLUS_HALF_F + wex : 0.f);
      if (r == 0) sm[S::MISC + 22] = ent;
    }
  };
  // ---- synchronisation between the two waves of a pair: a phase counter per wave in LDS.  The producer drains its LDS stores
  // (they complete in issue order) and raises its counter; the consumer polls the partner's counter, then reads.  A wave only
  // ever exchanges data with its partner before the dz1 barrier, so the other pairs need not be there yet.
  int* const pflag = reinterpret_cast<int*>(sm + S::MISC + 48);      // [8] one word per wave
  int pphase = 0;
#if ICRL_DIAG & 8
  auto pair_signal = [&]() { (void)pflag; (void)pphase; (void)partner; };
  auto pair_wait = [&]() {};
#elif ICRL_PAIR_FLAGS
  auto pair_signal = [&]() {
    ++pphase;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(pflag + w, pphase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto pair_wait = [&]() {
    while (__hip_atomic_load(pflag + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < pphase) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
  };
#else
  auto pair_signal = [&]() { (void)pflag; (void)pphase; (void)partner; lds_barrier(); };
  auto pair_wait = [&]() {};
#endif
  // lane 0 of wave ICRL_BOOK_WAVE keeps the running statistics of the role.  A HIGH wave (fh == 1): those have no head-weight
  // gradient to form and reach the norm barrier ~1 k cycles before the low waves, which is about what the book-keeping costs
  // (per-wave timers: on wave 3 it made that wave the last one at the barrier by ~750 cycles)
  const bool book = !(ICRL_DIAG & 1024) && tid == 64 * ICRL_BOOK_WAVE;
  // Its running sums live in LDS (MISC + 56..61) and are advanced with ds_add_f32, which needs no answer: as loop-carried registers of
  // one lane they were spilled to scratch at the 256-register limit (two scratch reloads + three stores per optimiser step on the
  // book-keeping wave; in-loop scratch instructions 12 -> 3; measured 8.91-8.93 against 8.94-8.95 us per step on the same box).  One
  // lane adds one value per step, so every sum is the same sequence of float32 additions as `st += x`.
  float* const acc_ent = sm + S::MISC + 56; float* const acc_pg = sm + S::MISC + 57; float* const acc_cf = sm + S::MISC + 58;
  float* const acc_vl = sm + S::MISC + 59; float* const acc_last = sm + S::MISC + 60; float* const acc_kl = sm + S::MISC + 61;
  if (tid >= 192 && tid < 192 + 6) sm[S::MISC + 56 + (tid - 192)] = 0.f;
  auto lds_add = [](float* p, float x) { __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  int steps_done = 0, early_stop_epoch = a.hp.n_epochs, status = 0;

  // ---- pipeline prologue
  int g_chunk = 0;
  int idx_next = chunk_idx(ld_chunk(1)), idx_nx2 = chunk_idx(ld_chunk(2));
  int2 pc_nx3 = ld_chunk(3);
  issue_rows(chunk_idx(ld_chunk(0)));
  int4 ps_next = ld_step(0), ps_nx2 = ld_step(1), ps_nx3 = ld_step(2);
  issue_stats(stat_idx(ps_next));
  int sidx_next = stat_idx(ps_nx2);
  if (tid == 0) sm[S::MISC + 14] = run_on_one_xcd(xch, role, 3) ? 1.f : 0.f;
  __syncthreads();                      // initial weights visible (refresh_gauss reads log_std)
  const bool xcd_local = __builtin_amdgcn_readfirstlane(__float_as_int(sm[S::MISC + 14])) != 0;
  refresh_gauss();
  int xcur = S::XT0;
  commit_rows(xcur);
  stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  __syncthreads();
  read_stats(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  const float inv_n_mb = 1.f / (float)((T * N + a.hp.batch_size - 1) / a.hp.batch_size);

  // ---- layer 1 of the chunk staged at `xbase`: own feature tiles t = 2 fh + tt -> h1c (registers) and the h1^T image
  const int b = 16 * rt + r;            // this lane's row of the chunk (all four q lanes share it)
  float* const pt = sm + (4 * q) * ST + b;     // + image + (16 t + i) ST: element [feature 16 t + 4 q + i][row b]
  f32x4 h1c[2];
  // TAILQ: the last K group holds 1..4 real observation components (obs 18: k = 16, 17).  An MFMA covers k = 16 js + 4 q + e for its
  // four lane groups q, so that group used to take one MFMA per component, each with three idle lane groups; here ONE MFMA takes the
  // components along q (k = 16 js + q): same products in the same order (the others are zeros), one MFMA and one operand fetch less
  // per component and tile.
  constexpr int JT = OBS / 16;
  constexpr bool TAILQ = ICRL_L1_TAILQ && OBS > 0 && OBS % 16 >= 1 && OBS % 16 <= 4 && JT < NT1;
  auto l1_forward = [&](int xbase) {
    float bx[NT1][4];                   // x[row b][k = 16 js + 4 q + e]
    const float* pb = sm + xbase + (4 * q) * ST + b;
#pragma unroll
    for (int js = 0; js < NT1; ++js)
#pragma unroll
      for (int e = 0; e < 4; ++e) bx[js][e] = ((OBS == 0 || 16 * js + e < OBS) && !(TAILQ && js == JT)) ? pb[(16 * js + e) * ST] : 0.f;
    const float bt = TAILQ ? sm[xbase + (16 * JT + q) * ST + b] : 0.f;      // x[row b][k = 16 JT + q]
    f32x4 z[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * fh + tt;
      const float* pa = sm + S::W1 + (16 * t + r) * SX + 4 * q;
      f32x4 aw[NT1];
#pragma unroll
      for (int js = 0; js < NT1; ++js)
        if (!(TAILQ && js >= JT)) aw[js] = lds128(pa + 16 * js);
      const float at = TAILQ ? sm[S::W1 + (16 * t + r) * SX + 16 * JT + q] : 0.f;
      z[tt] = lds128(sm + S::B1 + 16 * t + 4 * q);      // the bias is the accumulator's initial value
#pragma unroll
      for (int js = 0; js < NT1; ++js)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((OBS == 0 || 16 * js + e < OBS) && !(TAILQ && js >= JT)) z[tt] = MFMA_F32(aw[js][e], bx[js][e], z[tt]);     // (k >= obs: zero weights)
      if (TAILQ) z[tt] = MFMA_F32(at, bt, z[tt]);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int i = 0; i < 4; ++i) h1c[tt][i] = fast_tanh(z[tt][i]);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int i = 0; i < 4; ++i) pt[S::H1T + (16 * (2 * fh + tt) + i) * ST] = h1c[tt][i];
  };
#if ICRL_L1_AHEAD
  l1_forward(xcur);       // layer 1 of the first minibatch; from here on every step ends with layer 1 of the next one
  lds_barrier();
#endif

#if ICRL_HIGH_PRIO
  if (fh == 1) __builtin_amdgcn_s_setprio(ICRL_HIGH_PRIO);      // (A/B: the younger wave of a pair loses the issue arbitration to the older one)
#elif ICRL_LOW_PRIO
  if (fh == 0) __builtin_amdgcn_s_setprio(ICRL_LOW_PRIO);
#endif
  constexpr bool prof = PROF;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef ICRL_FINE_PROF
  unsigned long long fph[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  unsigned long long t_last = prof ? stamp() : 0ull;

  bool stop = false;
  for (int st = 0; st < n_steps && !stop; ++st) {
    const unsigned step = (unsigned)st + 1u;
    PlanStep ps;
    ps.step_size = __int_as_float(__builtin_amdgcn_readfirstlane(ps_next.x));
    ps.inv_bc2_sqrt = __int_as_float(__builtin_amdgcn_readfirstlane(ps_next.y));
    ps.nb_flags = __builtin_amdgcn_readfirstlane(ps_next.z);
    ps.perm_base = 0;
    ps_next = ps_nx2; ps_nx2 = ps_nx3;
    ps_nx3 = ld_step(st + 3 < n_steps + 2 ? st + 3 : n_steps + 1);
    const int nb = ps.nb_flags & NB_MASK;
    const float inv_nb = __builtin_amdgcn_rcpf((float)nb);
    const float c_mean_r = mean_r, c_mean_c = mean_c, c_istd_r = istd_r;
    const float cpol_nb = inv_nb * __builtin_amdgcn_rcpf(1.f + nu);
    issue_stats(sidx_next);
    sidx_next = stat_idx(ps_nx2);
    float mb_s0 = 0.f, mb_s1 = 0.f, mb_s2 = 0.f, mb_s3 = 0.f, mb_s4 = 0.f;

    const int n_chunks = (nb + RB - 1) / RB;
    for (int ch = 0; ch < n_chunks; ++ch, ++g_chunk) {
      const int nrows = (nb - ch * RB) < RB ? (nb - ch * RB) : RB;
      if (ch > 0) {
        if (S::XDB) xcur = xcur == S::XT0 ? S::XT1 : S::XT0;
        if (!EARLY_COMMIT) {
          commit_rows(xcur);
          lds_barrier();                    // a row is staged by threads of both waves of its pair
        }                                   // (EARLY_COMMIT: staged in front of the previous chunk's dz1 barrier, S5 and the chunk-end barrier passed since)
      }
      const bool valid = b < nrows;
      // ================= forward =================
      f32x4 h2c[2], outc;                   // own feature tiles t = 2 fh + tt
#if ICRL_L1_AHEAD
      if (ch > 0) l1_forward(xcur);         // (chunk 0: computed at the end of the previous step, under its Adam)
#else
      l1_forward(xcur);
#endif
      FSTAMP(0)   // L1
      // prefetch the next chunk's rows (random 72-byte pieces of the rollout buffer: several microseconds away)
      {
        const int idx_now = idx_next;
        idx_next = idx_nx2;
        idx_nx2 = chunk_idx(pc_nx3);        // (older loads first: what they wait for arrived a step ago)
        pc_nx3 = ld_chunk(g_chunk + 4);
        issue_rows(idx_now);
      }
      FSTAMP(1)   // row prefetch issue
#if ICRL_L1_AHEAD
      if (ch > 0) pair_signal();     // (P1) this wave's columns of h1^T are complete (chunk 0: the S7 barrier said so)
#else
      pair_signal();                 // (P1) this wave's columns of h1^T are complete
#endif
      FSTAMP(2)   // S1
      {
        // the own half of K (h1c, registers) needs nobody: its MFMAs run before the wait for the partner's half
        f32x4 z[2], awp[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * fh + tt;
          // (operand tiles addressed, never indexed, by the runtime half fh: a runtime-indexed register array goes to scratch)
          const float* pa = sm + S::W2 + (16 * t + r) * SH + 4 * q;
          f32x4 awo[2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) { awo[jj] = lds128(pa + 32 * fh + 16 * jj); awp[tt][jj] = lds128(pa + 32 * (1 - fh) + 16 * jj); }
          z[tt] = lds128(sm + S::B2 + 16 * t + 4 * q);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) z[tt] = MFMA_F32(awo[jj][e], h1c[jj][e], z[tt]);
        }
#if ICRL_L1_AHEAD
        if (ch > 0) pair_wait();     // the partner's columns of h1^T are complete
#else
        pair_wait();                 // the partner's columns of h1^T are complete
#endif
        float hpart[2][4];                  // the partner's half of h1 as B operand: features 16 js + 4 q + e, js = 2 (1 - fh) + jj
        const float* ph1 = sm + S::H1T + (4 * q) * ST + b + 32 * (1 - fh) * ST;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e) hpart[jj][e] = ph1[(16 * jj + e) * ST];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) z[tt] = MFMA_F32(awp[tt][jj][e], hpart[jj][e], z[tt]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 4; ++i) h2c[tt][i] = fast_tanh(z[tt][i]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::H2T + (16 * (2 * fh + tt) + i) * ST] = h2c[tt][i];
      }
      FSTAMP(3)   // L2
      {  // head, split over K: this wave's 32 features; partial tiles exchanged through LDS
        const float* pa = sm + S::WH + r * SH + 4 * q + 32 * fh;
        f32x4 acc = low ? lds128(sm + S::BH + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f32x4 aw = lds128(pa + 16 * jj);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = MFMA_F32(aw[e], h2c[jj][e], acc);
        }
        // the pair's partial tiles live in the pair's OWN columns (16 rt ..) of the dz1^T image: rows 16 fh + lane / 4
        float* const hpx = sm + S::HPX + (lane >> 2) * ST + 16 * rt + 4 * (lane & 3);
        *reinterpret_cast<f32x4*>(hpx + 16 * fh * ST) = acc;
        FSTAMP(4)   // head MFMA + partial store
        pair_signal(); pair_wait();  // (P3) both partial tiles stored
        FSTAMP(5)   // S3
        const f32x4 p0 = lds128(hpx), p1 = lds128(hpx + 16 * ST);
#pragma unroll
        for (int i = 0; i < 4; ++i) outc[i] = p0[i] + p1[i];
      }
      STAMP(0)   // forward
      // ============ loss + d loss / d head output (both waves of a pair: identical values) ============
      f32x4 dout = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ICRL_DIAG & 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dout[i] = outc[i] * 1e-6f;
      } else {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (role == 0) {
          const int ngp = ICRL_HEAD_PERM ? (A + 3) >> 2 : 4;      // groups of four outputs that hold real ones
          float lp = 0.f, ent = 0.f;
          f32x4 g1 = f32x4{0.f, 0.f, 0.f, 0.f}, g2 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (DISC) {
            float lg[4], zmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) { lg[i] = (out_of(i) < A) ? outc[i] : -INFINITY; zmax = fmaxf(zmax, lg[i]); }
            zmax = xor16_max(xor32_max(zmax));
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) se += (out_of(i) < A) ? expf(lg[i] - zmax) : 0.f;
            se = quad_rows_sum(se);
            const float lse = zmax + logf(se);
            const int act = (int)sm[S::ACT + b * SA];
            float pr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = out_of(i);
              lg[i] = k < A ? lg[i] - lse : 0.f;
              pr[i] = k < A ? expf(lg[i]) : 0.f;
              lp += (k == act) ? lg[i] : 0.f;
              ent -= pr[i] * lg[i];
            }
            lp = quad_rows_sum(lp);
            ent = quad_rows_sum(ent);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = out_of(i);
              g1[i] = k < A ? ((k == act ? 1.f : 0.f) - pr[i]) : 0.f;
              g2[i] = k < A ? pr[i] * (lg[i] + ent) : 0.f;
            }
          } else {
            const f32x4 actv = lds128(sm + S::ACT + b * SA + 4 * q);
            const f32x4 iv = lds128(sm + S::GAU + 4 * q), hiv = lds128(sm + S::GAU + 16 + 4 * q), lsd = lds128(sm + S::GAU + 32 + 4 * q);
            // elements i with 4 i >= A are head padding in every lane group (out_of): all their terms are zeros, skipped
            auto elem = [&](int i) {
              const float dd = actv[i] - outc[i];
              lp += -(dd * dd) * hiv[i] - lsd[i];
              g1[i] = dd * iv[i];
              g2[i] = (out_of(i) < A) ? (dd * dd) * iv[i] - 1.f : 0.f;
            };
            elem(0);
            if (ngp > 1) { elem(1); if (ngp > 2) { elem(2); elem(3); } }
            lp = quad_rows_sum(lp);
          }
          const float old_lp = sm[S::OLP + b];
          const float ratio = __expf(lp - old_lp);
          const float Ar = (sm[S::ADR + b] - c_mean_r) * c_istd_r;
          const float Ac = sm[S::ADC + b] - c_mean_c;
          const float s1 = Ar * ratio;
          const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
          const float s2 = Ar * rc;
          const float gsel = (s1 <= s2) ? Ar : 0.f;
          const float dlp = valid ? cpol_nb * (-gsel + nu * Ac) * ratio : 0.f;
          if (DISC) {
            const float dent = valid ? ent_coef * inv_nb : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) dout[i] = dlp * g1[i] + dent * g2[i];
          } else {
            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};       // d log_std: sum over this tile's 16 rows, per element i (output out_of(i))
            auto dls = [&](int i) { dout[i] = dlp * g1[i]; t[i] = row_sum(dlp * g2[i]); };
            dls(0);
            if (ngp > 1) { dls(1); if (ngp > 2) { dls(2); dls(3); } }
            if (low && r == 0) *reinterpret_cast<f32x4*>(sm + S::PLS + 16 * rt + 4 * q) = t;
          }
          const bool cnt = valid && q == 0;
          v0 = cnt ? fminf(s1, s2) : 0.f; v1 = cnt ? Ac * ratio : 0.f; v2 = (cnt && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
          v3 = cnt ? old_lp - lp : 0.f; v4 = cnt ? ent : 0.f;
        } else {
          const float v = quad_rows_sum(q == 0 ? outc[0] : 0.f);
          const float R = sm[S::ADR + b];
          float vp = v, pass = 1.f;
          if (vclip >= 0.f) {
            const float old = sm[S::OLP + b];
            const float dv = v - old;
            vp = old + fminf(fmaxf(dv, -vclip), vclip);
            pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
          }
          const float e = vp - R;
          const float d0 = valid ? vcoef * 2.f * e * inv_nb * pass : 0.f;
          dout[0] = q == 0 ? d0 : 0.f;
          v0 = (valid && q == 0) ? e * e : 0.f;
        }
        if (low) {     // one wave of the pair reports the tile's statistics
          v0 = row_sum(v0); v1 = row_sum(v1); v2 = row_sum(v2); v3 = row_sum(v3);
          if (DISC) v4 = row_sum(v4);
          if (lane == 0) { float* pst = sm + S::PST + 8 * rt; pst[0] = v0; pst[1] = v1; pst[2] = v2; pst[3] = v3; pst[4] = v4; }
        }
      }
      STAMP(1)   // loss
      FSTAMP(6)   // partial read + loss
      // ================= backward of the activations =================
      f32x4 dz2c[2], dz1c[2];
      {  // dH2^T = Wh^T . dOut^T for the own feature tiles: A = WHT[j = 16 t + r][position 4 q + e] (K = 16 head positions);
         // MFMA e covers the outputs 4 e .. 4 e + 3 (out_of): those with 4 e >= n_out would multiply zeros
        const int ng = ICRL_HEAD_PERM ? (n_out + 3) >> 2 : 4;
        f32x4 aw[2], acc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * fh + tt;
          if (S::WHTC) aw[tt] = lds128(sm + S::WHT + (16 * t + r) * SA + 4 * q);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) aw[tt][e] = sm[S::WH + (4 * q + e) * SH + 16 * t + r];
          }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = MFMA_F32(aw[tt][0], dout[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
        if (ng > 1) {
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) acc[tt] = MFMA_F32(aw[tt][1], dout[1], acc[tt]);
          if (ng > 2) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { acc[tt] = MFMA_F32(aw[tt][2], dout[2], acc[tt]); acc[tt] = MFMA_F32(aw[tt][3], dout[3], acc[tt]); }
          }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * fh + tt;
#pragma unroll
          for (int i = 0; i < 4; ++i) dz2c[tt][i] = fmaf(-(h2c[tt][i] * h2c[tt][i]), acc[tt][i], acc[tt][i]);   // acc (1 - h2^2)
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::DZ2T + (16 * t + i) * ST] = dz2c[tt][i];
        }
        if (low) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::DOT + i * ST] = dout[i];
        }
      }
      FSTAMP(7)   // dH2
      pair_signal();               // (P4) this wave's columns of dz2^T complete
      FSTAMP(8)   // S4
      {  // dH1^T = W2^T . dz2^T: A = W2T[k = 16 t + r][j = 16 js + 4 q + e]; own half of K (dz2c, registers) before the wait for
         // the partner's half, which comes from the image
        f32x4 acc[2], awp[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * fh + tt;
          f32x4 awo[2];
          if (S::W2TC) {
            const float* pa = sm + S::W2T + (16 * t + r) * SH + 4 * q;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) { awo[jj] = lds128(pa + 32 * fh + 16 * jj); awp[tt][jj] = lds128(pa + 32 * (1 - fh) + 16 * jj); }
          } else {
            const float* pa = sm + S::W2 + (4 * q) * SH + 16 * t + r;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                awo[jj][e] = pa[(32 * fh + 16 * jj + e) * SH];
                awp[tt][jj][e] = pa[(32 * (1 - fh) + 16 * jj + e) * SH];
              }
          }
          acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[tt] = MFMA_F32(awo[jj][e], dz2c[jj][e], acc[tt]);
        }
        pair_wait();               // the partner's columns of dz2^T complete
        float dp[2][4];
        const float* pz = sm + S::DZ2T + (4 * q) * ST + b + 32 * (1 - fh) * ST;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e) dp[jj][e] = pz[(16 * jj + e) * ST];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[tt] = MFMA_F32(awp[tt][jj][e], dp[jj][e], acc[tt]);
#pragma unroll
          for (int i = 0; i < 4; ++i) dz1c[tt][i] = fmaf(-(h1c[tt][i] * h1c[tt][i]), acc[tt][i], acc[tt][i]);
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::DZ1T + (16 * (2 * fh + tt) + i) * ST] = dz1c[tt][i];
      STAMP(2)   // activation backward
      FSTAMP(9)   // dH1 + dz1 store
      if (EARLY_COMMIT) {
        // ICRL_EARLY_COMMIT (measured and rejected: 8.81 us per step against 8.46; off — the dz1 barrier waits for the HIGH waves, and the
        // staging lands on their path).  The prefetched rows (the next chunk's, or the next minibatch's first chunk) are staged HERE, in front of the dz1 barrier, instead of
        // behind the norm publish: the other X^T buffer is free during the whole chunk, the per-row action / scalar images belong to this
        // pair alone and its loss tail is over (the partner's dz2 hand-off has been passed).  At the end of the step the low waves are the
        // critical ones (they carry dWh) and the staging sat on their path; here it sits where they wait for the dz1 barrier.
        commit_rows(xcur == S::XT0 ? S::XT1 : S::XT0);
        if (ch + 1 == n_chunks) stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
      }
      if (ch == 0) {   // gradient accumulators start their life here
#pragma unroll
        for (int cc = 0; cc < NW1; ++cc) gW1r[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        gW2r[0] = gW2r[1] = gWhr = f32x4{0.f, 0.f, 0.f, 0.f};
        gb1r = 0.f; gb2r = 0.f; gex = 0.f; gt[0] = gt[1] = 0.f;
      }
      if (!(ICRL_DIAG & 1)) lds_barrier();  // (S5) every pair's columns of h1^T, h2^T, dz1^T, dz2^T, dOut^T (and the loss partials) are complete
      FSTAMP(12)  // S5
      // ================= weight gradients (K = the 64 rows) =================
      {  // dW2 rows 16 rt.., column tiles 2 fh, 2 fh + 1
        f32x4 az[4];   // dz2^T[j = 16 rt + r][rows 16 js + 4 q + e]
        const float* pa = sm + S::DZ2T + (16 * rt + r) * ST + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) az[js] = lds128(pa + 16 * js);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const float* pb = sm + S::H1T + (16 * (2 * fh + cc) + r) * ST + 4 * q;
          f32x4 bh[4];
#pragma unroll
          for (int js = 0; js < 4; ++js) bh[js] = lds128(pb + 16 * js);
          OPERANDS_FIRST();
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) gW2r[cc] = MFMA_F32(az[js][e], bh[js][e], gW2r[cc]);
        }
        if (low) {
          float s = 0.f;     // d b2[16 rt + r] = sum over the rows
#pragma unroll
          for (int js = 0; js < 4; ++js) s += (az[js][0] + az[js][1]) + (az[js][2] + az[js][3]);
          gb2r += quad_rows_sum(s);
        }
      }
      FSTAMP(10)  // dW2
      if (own_wh) {   // dWh columns 16 rt..: A = dOut^T[o = r][rows], B = h2^T[j = 16 rt + r][rows]
        f32x4 ao[4], bh[4];
        const float* pa = sm + S::DOT + r * ST + 4 * q;
        const float* pb = sm + S::H2T + (16 * rt + r) * ST + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) { ao[js] = lds128(pa + 16 * js); bh[js] = lds128(pb + 16 * js); }
        OPERANDS_FIRST();
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int js = 0; js < 4; ++js)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[js & 1] = MFMA_F32(ao[js][e], bh[js][e], acc[js & 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) gWhr[i] += acc[0][i] + acc[1][i];
        float s = 0.f;      // head bias (wave 0 keeps it): row sums of dOut^T
#pragma unroll
        for (int js = 0; js < 4; ++js) s += (ao[js][0] + ao[js][1]) + (ao[js][2] + ao[js][3]);
        s = quad_rows_sum(s);
        // log_std's wave (Gaussian policy): d log_std r = sum of the four per-tile partials
        const float sl = (sm[S::PLS + r] + sm[S::PLS + 16 + r]) + (sm[S::PLS + 32 + r] + sm[S::PLS + 48 + r]);
        gex += w == W_BH ? s : ((!DISC && role == 0 && w == W_LS) ? sl : 0.f);
      }
      FSTAMP(11)  // dWh
      {  // dW1 rows 16 rt.., observation tiles NW1 fh ..
        f32x4 az[4];   // dz1^T[j = 16 rt + r][rows]
        const float* pa = sm + S::DZ1T + (16 * rt + r) * ST + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) az[js] = lds128(pa + 16 * js);
        if (TAILW && !low) {      // the tail columns k = 16 + c as dot products over this lane's 16 rows, summed over the lane groups
#pragma unroll
          for (int c = 0; c < NTAIL; ++c) {
            const float* px_ = sm + xcur + (16 + c) * ST + 4 * q;      // x^T[16 + c][rows 16 js + 4 q + e] (the same for every r)
            float s_ = 0.f;
#pragma unroll
            for (int js = 0; js < 4; ++js) {
              const f32x4 xr = lds128(px_ + 16 * js);
#pragma unroll
              for (int e = 0; e < 4; ++e) s_ = fmaf(az[js][e], xr[e], s_);
            }
            gt[c] += quad_rows_sum(s_);
          }
        }
#pragma unroll
        for (int cc = 0; cc < NW1; ++cc) {
          if (!own_w1_tile) break;
          const float* pb = sm + xcur + (16 * (NW1 * fh + cc) + r) * ST + 4 * q;     // x^T[k][rows 16 js + 4 q + e]
          f32x4 bx[4];
#pragma unroll
          for (int js = 0; js < 4; ++js) bx[js] = lds128(pb + 16 * js);
          OPERANDS_FIRST();
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) gW1r[cc] = MFMA_F32(az[js][e], bx[js][e], gW1r[cc]);
          // observation pad columns (k >= obs): X holds unmasked fill there; their weights, gradients and moments stay 0
          if (ch + 1 == n_chunks && 16 * (NW1 * fh + cc) + r >= O) gW1r[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (low) {
          float s = 0.f;
#pragma unroll
          for (int js = 0; js < 4; ++js) s += (az[js][0] + az[js][1]) + (az[js][2] + az[js][3]);
          gb1r += quad_rows_sum(s);
        }
      }
      if (book) {
        mb_s0 += (sm[S::PST + 0] + sm[S::PST + 8]) + (sm[S::PST + 16] + sm[S::PST + 24]);
        mb_s1 += (sm[S::PST + 1] + sm[S::PST + 9]) + (sm[S::PST + 17] + sm[S::PST + 25]);
        mb_s2 += (sm[S::PST + 2] + sm[S::PST + 10]) + (sm[S::PST + 18] + sm[S::PST + 26]);
        mb_s3 += (sm[S::PST + 3] + sm[S::PST + 11]) + (sm[S::PST + 19] + sm[S::PST + 27]);
        if (DISC) mb_s4 += (sm[S::PST + 4] + sm[S::PST + 12]) + (sm[S::PST + 20] + sm[S::PST + 28]);
      }
      if (ch + 1 < n_chunks || !S::XDB) lds_barrier();  // chunk buffers free (single X buffer: also before it is restaged)
      STAMP(3)   // weight gradients
      FSTAMP(13)  // dW1
    }  // chunks

    // entropy term of the Gaussian policy loss: d(ent_coef * -mean(H)) / d log_std = -ent_coef
    if (!DISC && role == 0 && w == W_LS && ro < A) gex += -ent_coef;

    // ================= global gradient norm: this wave's partial sum of squares -> its own 8-byte granule =================
    float ss = 0.f;
    if (ICRL_DIAG & 256) ss = 1.f + gW1r[0][0] * 1e-9f; else {
#pragma unroll
    for (int cc = 0; cc < NW1; ++cc)
#pragma unroll
      for (int i = 0; i < 4; ++i) ss = fmaf(gW1r[cc][i], gW1r[cc][i], ss);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int i = 0; i < 4; ++i) ss = fmaf(gW2r[cc][i], gW2r[cc][i], ss);
    {
      float sb = 0.f;      // per-row entries (replicated over the lane groups: counted on q == 0)
      if (own_wh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ss = fmaf(gWhr[i], gWhr[i], ss);
        sb = ex_g >= 0 ? gex * gex : 0.f;
      }
      if (low) sb = fmaf(gb1r, gb1r, gb2r * gb2r) + sb;
      if (TAILW && !low) sb = fmaf(gt[0], gt[0], gt[1] * gt[1]) + sb;
      ss += q == 0 ? sb : 0.f;
    }
    ss = wave_sum_fast(ss);
    }
    if (lane == 0) {
      bool want_stop = false;
      float mean_kl = 0.f;
      const bool last_mb = (ps.nb_flags >> NB_LAST) & 1;
      const int epoch = ps.nb_flags >> NB_EPOCH;
      if (book && role == 0) {   // the early-stop decision rides on the granule of the policy workgroup's book-keeping wave
        if ((ps.nb_flags >> NB_FIRST) & 1) *acc_kl = 0.f;
        lds_add(acc_kl, mb_s3 * inv_nb);
        if (last_mb) {
          mean_kl = __hip_atomic_load(acc_kl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * inv_n_mb;
          const TrainArgs* k_ = KARGS();
          if (k_->hp.use_target_kl && mean_kl > 1.5f * k_->hp.target_kl) { want_stop = true; early_stop_epoch = epoch; }
        }
      }
      const unsigned tag = step | (want_stop ? 0x80000000u : 0u);
      // (xcd_local: the three workgroups share an XCD — the line stays in its L2 for the others' polls; ppo_common.h)
      if (xcd_local) __hip_atomic_store(xch + (step & 1) * 32 + role * 8 + w, ((u64)tag << 32) | (u64)__float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(xch + (step & 1) * 32 + role * 8 + w, ((u64)tag << 32) | (u64)__float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // this workgroup reads its OWN eight partials from LDS (same floats, same summation order as everybody else's view of
      // them): the role that publishes last — the policy, the critical path — does not wait for its own stores to come back
      // through the memory system (~870 cycles)
#if ICRL_OWN_LDS
      sm[S::MISC + 24 + role * 8 + w] = ss;
      if (book && role == 0) sm[S::MISC + 12] = want_stop ? 1.f : 0.f;
#endif
      if (book) {
        ++steps_done;
        if (role == 0) {
          float ent = 0.f;
          if (DISC) ent = mb_s4 * inv_nb;
          else ent = sm[S::MISC + 22];
          const float entropy_loss = -ent;
          const float pl = (-(mb_s0 * inv_nb) + nu * (mb_s1 * inv_nb)) * __builtin_amdgcn_rcpf(1.f + nu);
          lds_add(acc_ent, entropy_loss); lds_add(acc_pg, pl); lds_add(acc_cf, mb_s2 * inv_nb);
          *acc_last = pl + ent_coef * entropy_loss;
          if (last_mb) { float* stats = KARGS()->stats; stats[32 + epoch] = mean_kl; stats[7] = mean_kl; }
        } else {
          const float vl = mb_s0 * inv_nb;
          lds_add(acc_vl, vl);
          *acc_last = vl;
        }
      }
    }
    STAMP(4)   // gradient norm + publish
    FSTAMP(14)  // norm + publish
    // ---- while the granules travel: stage the next minibatch (rows -> the other X^T buffer, advantage statistics)
    const int xnext = S::XDB ? (xcur == S::XT0 ? S::XT1 : S::XT0) : xcur;
    const int nb_next = __builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK;
    // granule (role rr, wave ww) = slot 8 rr + ww; a workgroup polls the other two roles' sixteen.  The FIRST look is issued before
    // the staging and read behind it: the policy workgroup publishes last, so the critics' granules are in memory by now and the
    // look's trip through the memory system (~0.9 k cycles) runs under the staging instead of behind it.
    const bool poller = tid < 24 && (!ICRL_OWN_LDS || (tid >> 3) != role);
    const u64* const slot = xch + (step & 1) * 32 + (tid < 24 ? tid : 0);
#if ICRL_EARLY_POLL
    u64 v_first = 0;
    if (poller) v_first = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (!EARLY_COMMIT) {
      commit_rows(xnext);
      stats_partials(nb_next);
    }
    xcur = xnext;
    FSTAMP(15)  // staging
    if (poller) {
      u64 v = 0;
      int spins = 0;
      bool ok = false;
#if ICRL_EARLY_POLL
      v = v_first;
      ok = (unsigned)((v >> 32) & 0x7fffffffu) == step;
#endif
      if (ICRL_DIAG & 256) ok = true;
      while (!ok && spins < (1 << 24)) {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)((v >> 32) & 0x7fffffffu) == step) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
        ++spins;
      }
      sm[S::MISC + 24 + tid] = __uint_as_float((unsigned)(v & 0xffffffffu));
      if (tid == ICRL_BOOK_WAVE) sm[S::MISC + 12] = (v >> 63) ? 1.f : 0.f;     // the granule of the policy's book-keeping wave carries the stop flag
      if (!ok) sm[S::MISC + 13] = 1.f;
    }
    if (!(ICRL_DIAG & 2)) lds_barrier();   // (S6) norm partials, next minibatch and its statistics visible
    STAMP(5)   // staging + granule wait
    FSTAMP(16)  // poll + S6
    float total = 0.f;
    {
#pragma unroll
      for (int g = 0; g < 6; ++g) {     // fixed order: every wave of every role forms the same total
        const f32x4 n = lds128(sm + S::MISC + 24 + 4 * g);
        total += (n[0] + n[1]) + (n[2] + n[3]);
      }
      const f32x4 fl = lds128(sm + S::MISC + 12);
      stop = fl[0] != 0.f;
      if (fl[1] != 0.f) { status = 1; stop = true; }
    }
    total = __builtin_amdgcn_sqrtf(total);
    float coef = max_grad_norm * __builtin_amdgcn_rcpf(total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    read_stats(nb_next > 0 ? nb_next : 2);

    // ================= Adam (torch.optim.Adam, single-tensor form) on the wave's own elements =================
    {
      const float step_size = ps.step_size, inv_bc2_sqrt = ps.inv_bc2_sqrt;
      const float epsf = adam_epsf;
      const float omw1 = 1.f - w1, b2f_ = adam_b2f;
      const float cw1 = coef * w1, c2w2 = (coef * coef) * w2;
      auto adam4 = [&](const f32x4& g, f32x4& m, f32x4& v, f32x4& p) {   // stage by stage: four independent chains
        if (ICRL_DIAG & 32) { p[0] += g[0] * step_size; return; }
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = fmaf(cw1, g[i], omw1 * m[i]); v[i] = fmaf(c2w2, g[i] * g[i], b2f_ * v[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = fmaf(__builtin_amdgcn_sqrtf(v[i]), inv_bc2_sqrt, epsf);   // v_sqrt_f32 / v_rcp_f32: 1 ulp each
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = fmaf(-step_size, m[i] * __builtin_amdgcn_rcpf(d[i]), p[i]);
      };
      // pad elements (k >= obs, o >= n_out) have g = m = v = p = 0 and stay 0: no masks needed
      // Part 1: what layer 1 reads (W1, b1; the low waves' bias block carries b2 and the head bias / log_std along)
      if (own_w1_tile) {
#pragma unroll
        for (int cc = 0; cc < NW1; ++cc) { f32x4 p_ = load_own_w1(cc); adam4(gW1r[cc], mW1[cc], vW1[cc], p_); store_w1(cc, p_); }
      }
      if (TAILW) {      // per-row entries, lane q == 0 stores — low: {b1, b2}, high: {tail column 0, tail column 1}; the head bias | log_std rides with its owner
        const bool exo = own_wh && ex_g >= 0;
        if (low) {
          f32x4 g_ = f32x4{gb1r, gb2r, exo ? gex : 0.f, 0.f}, p_ = f32x4{sm[S::B1 + jb], sm[S::B2 + jb], own_wh ? sm[ex_s] : 0.f, 0.f};
          f32x4 m_ = f32x4{mb1, mb2, mex, 0.f}, v_ = f32x4{vb1, vb2, vex, 0.f};
          adam4(g_, m_, v_, p_);
          mb1 = m_[0]; mb2 = m_[1]; mex = m_[2]; vb1 = v_[0]; vb2 = v_[1]; vex = v_[2];
          if (q == 0) { sm[S::B1 + jb] = p_[0]; sm[S::B2 + jb] = p_[1]; if (own_wh) sm[ex_s] = p_[2]; }
        } else {
          float* const pw = sm + S::W1 + jb * SX + 16;
          f32x4 g_ = f32x4{gt[0], gt[1], exo ? gex : 0.f, 0.f}, p_ = f32x4{pw[0], NTAIL > 1 ? pw[1] : 0.f, own_wh ? sm[ex_s] : 0.f, 0.f};
          f32x4 m_ = f32x4{mt[0], mt[1], mex, 0.f}, v_ = f32x4{vt[0], vt[1], vex, 0.f};
          adam4(g_, m_, v_, p_);
          mt[0] = m_[0]; mt[1] = m_[1]; mex = m_[2]; vt[0] = v_[0]; vt[1] = v_[1]; vex = v_[2];
          if (q == 0) { pw[0] = p_[0]; if (NTAIL > 1) pw[1] = p_[1]; if (own_wh) sm[ex_s] = p_[2]; }
        }
      } else if (low) {
        f32x4 g_ = f32x4{gb1r, gb2r, ex_g >= 0 ? gex : 0.f, 0.f}, p_ = f32x4{sm[S::B1 + jb], sm[S::B2 + jb], sm[ex_s], 0.f};
        f32x4 m_ = f32x4{mb1, mb2, mex, 0.f}, v_ = f32x4{vb1, vb2, vex, 0.f};
        adam4(g_, m_, v_, p_);     // identical arithmetic in the four q lanes, lane q == 0 stores
        mb1 = m_[0]; mb2 = m_[1]; mex = m_[2]; vb1 = v_[0]; vb2 = v_[1]; vex = v_[2];
        if (q == 0) { sm[S::B1 + jb] = p_[0]; sm[S::B2 + jb] = p_[1]; sm[ex_s] = p_[2]; }
      }
      // Part 2: W2, head weights, the Gaussian head's constants
      auto adam_rest = [&]() {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { f32x4 p_ = load_own_w2(cc); adam4(gW2r[cc], mW2[cc], vW2[cc], p_); store_w2(cc, p_); }
        if (own_wh) {
          { f32x4 p_ = load_own_wh(); adam4(gWhr, mWh, vWh, p_); store_wh(p_); }
          __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the log_std store has landed before refresh_gauss re-reads it
          refresh_gauss();
        }
      };
#if ICRL_L1_AHEAD
      // Layer 1 of the NEXT minibatch (staged above) needs W1 and b1 only: the two waves of a SIMD run it and the rest of Adam in
      // opposite orders, so one wave's MFMAs / LDS latencies meet the other's VALU work instead of its twin
      lds_barrier();   // (S7a) W1, b1 visible
      FSTAMP(19)
      if (low) { l1_forward(xcur); adam_rest(); } else { adam_rest(); l1_forward(xcur); }
#else
      adam_rest();
#endif
    }
    FSTAMP(17)  // Adam
    if (!(ICRL_DIAG & 4)) lds_barrier();   // (S7) updated weights visible
    STAMP(6)   // Adam
    FSTAMP(18)  // S7
  }  // optimiser steps

  __syncthreads();
  // ---- write back weights, moments, statistics
  {
  const TrainArgs* kw = ka;
  asm volatile("" : "+s"(kw));
  const TrainArgs& a = *kw;
  const PolLayout& L = a.L;
  const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
  const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
  if (own_w1_tile) {
#pragma unroll
    for (int cc = 0; cc < NW1; ++cc) {
      const f32x4 pv = load_own_w1(cc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * rt + 4 * q + i, k = 16 * (NW1 * fh + cc) + r;
        if (k < O) { a.params[gW1 + j * O + k] = pv[i]; a.exp_avg[gW1 + j * O + k] = mW1[cc][i]; a.exp_avg_sq[gW1 + j * O + k] = vW1[cc][i]; }
      }
    }
  } else if (q == 0) {
#pragma unroll
    for (int c = 0; c < NTAIL; ++c) {
      const int e = gW1 + jb * O + 16 + c;
      a.params[e] = sm[S::W1 + jb * SX + 16 + c]; a.exp_avg[e] = mt[c]; a.exp_avg_sq[e] = vt[c];
    }
  }
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const f32x4 pv = load_own_w2(cc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * rt + 4 * q + i, k = 16 * (2 * fh + cc) + r;
      a.params[gW2 + j * HD + k] = pv[i];
      a.exp_avg[gW2 + j * HD + k] = mW2[cc][i];
      a.exp_avg_sq[gW2 + j * HD + k] = vW2[cc][i];
    }
  }
  if (own_wh) {
    const f32x4 pv = load_own_wh();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = out_of(i), j = 16 * rt + r;
      if (o < n_out) { a.params[gWh + o * HD + j] = pv[i]; a.exp_avg[gWh + o * HD + j] = mWh[i]; a.exp_avg_sq[gWh + o * HD + j] = vWh[i]; }
    }
    if (q == 0 && ex_g >= 0) { a.params[ex_g] = sm[ex_s]; a.exp_avg[ex_g] = mex; a.exp_avg_sq[ex_g] = vex; }
  }
  if (low && q == 0) {
    a.params[gb1 + jb] = sm[S::B1 + jb]; a.exp_avg[gb1 + jb] = mb1; a.exp_avg_sq[gb1 + jb] = vb1;
    a.params[gb2 + jb] = sm[S::B2 + jb]; a.exp_avg[gb2 + jb] = mb2; a.exp_avg_sq[gb2 + jb] = vb2;
  }
#ifdef ICRL_FINE_PROF
  // which wave reports: hp._pad bits 8..10 = wave, bits 12..13 = role (tools/train_only.py FINE=1)
  if (tid == 64 * ((a.hp._pad >> 8) & 7) && prof && role == ((a.hp._pad >> 12) & 3))
    for (int k = 0; k < 20; ++k) a.stats[12 + k] = (float)((double)fph[k] / (double)(a.n_steps > 0 ? a.n_steps : 1));
#endif
  if (tid == 0 && prof) {
#ifdef ICRL_FINE_PROF
    for (int k = 0; k < 0; ++k) {
#else
    for (int k = 0; k < 7; ++k) {
#endif
      const int slot = 12 + 7 * role + k;
      if (slot < 32) a.stats[slot] = (float)((double)ph[k] / (double)(a.n_steps > 0 ? a.n_steps : 1));   // (full runs only)
    }
  }
  if (book) {
    if (role == 0) {
      a.stats[0] = (float)early_stop_epoch;
      a.stats[1] = (float)steps_done;
      a.stats[2] = *acc_ent; a.stats[3] = *acc_pg; a.stats[6] = *acc_cf;
      a.stats[8] = *acc_last;
      a.stats[11] = (float)status;
      a.adam_t[0] = t0 + steps_done;
    } else if (role == 1) {
      a.stats[4] = *acc_vl; a.stats[9] = *acc_last;
    } else {
      a.stats[5] = *acc_vl; a.stats[10] = *acc_last;
    }
  }
  }
}

template <int NT1, bool DISC, int OBS, bool PROF>
__global__ void __launch_bounds__(TH8) ppo_train_pairs_kernel(TrainArgs a, int packed) {
  int run = 0, role = (int)blockIdx.x;
  if (packed && !packed_slot(3, 1, run, role)) return;
  ppo_train_pairs_body<NT1, DISC, OBS, PROF>(a, (const TrainArgs*)__builtin_amdgcn_kernarg_segment_ptr(), role);
}

// several independent runs in ONE launch: the packed 1-D grid of ppo_common.h (a run's workgroups on one XCD), or grid (3, n_runs) with
// run = blockIdx.y when that many workgroups are not resident at once; the argument blocks live in device memory
template <int NT1, bool DISC, int OBS, bool PROF>
__global__ void __launch_bounds__(TH8) ppo_train_pairs_batch_kernel(const TrainArgs* __restrict__ runs, int n_runs, int packed) {
  int run = (int)blockIdx.y, role = (int)blockIdx.x;      // run-major layout: grid (3, n_runs)
  if (packed && !packed_slot(3, n_runs, run, role)) return;
  const TrainArgs* const ka = as_global(runs + run);
  ppo_train_pairs_body<NT1, DISC, OBS, PROF>(*ka, ka, role);
}

// one: single-run launch (argument block by value) | d_args: n_runs blocks in device memory
template <int NT1, bool DISC, int OBS, bool PROF>
static int launch_pairs_p(const TrainArgs* one, const TrainArgs* d_args, int n_runs, hipStream_t s) {
  static_assert(SmemP<NT1>::TOTAL * sizeof(float) <= 160 * 1024, "LDS budget");
  const size_t bytes = ICRL_STATIC_LDS ? 0 : (size_t)SmemP<NT1>::TOTAL * sizeof(float);
  if (one != nullptr) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_train_pairs_kernel<NT1, DISC, OBS, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    TrainArgs arg = *one;
    return launch_update_single(ppo_train_pairs_kernel<NT1, DISC, OBS, PROF>, 3, dim3(TH8), bytes, s, arg);
  } else {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_train_pairs_batch_kernel<NT1, DISC, OBS, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    const int pg = packed_grid(3, n_runs);      // the runs' workgroups on one XCD each (ppo_common.h) when the whole grid is resident at once
    hipLaunchKernelGGL((ppo_train_pairs_batch_kernel<NT1, DISC, OBS, PROF>), pg ? dim3(pg) : dim3(3, n_runs), dim3(TH8), bytes, s, d_args, n_runs, pg ? 1 : 0);
  }
  return (int)hipGetLastError();
}

template <bool PROF>
static int dispatch_pairs_p(const TrainArgs* one, const TrainArgs* d_args, int n_runs, int obs, int nt1, bool discrete, hipStream_t s) {
  if (nt1 <= 2) {
    if (!discrete && obs == 18) return launch_pairs_p<2, false, 18, PROF>(one, d_args, n_runs, s);      // HCWithPos (BASELINE configs[1], [3])
    if (discrete && obs == 1) return launch_pairs_p<2, true, 1, PROF>(one, d_args, n_runs, s);          // LapGridWorld (configs[0])
    return discrete ? launch_pairs_p<2, true, 0, PROF>(one, d_args, n_runs, s) : launch_pairs_p<2, false, 0, PROF>(one, d_args, n_runs, s);
  }
  if (nt1 <= 4) return discrete ? launch_pairs_p<4, true, 0, PROF>(one, d_args, n_runs, s) : launch_pairs_p<4, false, 0, PROF>(one, d_args, n_runs, s);
  return fail("update (wave pairs): obs_dim tiles %d > 4", nt1);
}
static int dispatch_pairs(const TrainArgs* one, const TrainArgs* d_args, int n_runs, int obs, int nt1, bool discrete, bool prof, hipStream_t s) {
  return prof ? dispatch_pairs_p<true>(one, d_args, n_runs, obs, nt1, discrete, s) : dispatch_pairs_p<false>(one, d_args, n_runs, obs, nt1, discrete, s);
}

int launch_train_pairs_batch(const TrainArgs* d_args, int n_runs, int obs, int nt1, bool discrete, bool prof, hipStream_t s) {
  return dispatch_pairs(nullptr, d_args, n_runs, obs, nt1, discrete, prof, s);
}

int launch_train_pairs(const TrainArgs& a, int nt1, bool discrete, hipStream_t s) {
  return dispatch_pairs(&a, nullptr, 1, a.L.O, nt1, discrete, (a.hp._pad & 1) != 0, s);
}

}  // namespace icrl
