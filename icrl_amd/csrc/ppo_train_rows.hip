// PPO-Lagrangian update, row-owning-wave variant (obs_dim <= 128) — gfx950.
//
// ref: stable_baselines3/ppo_lag/ppo_lag.py:196-299, common/buffers.py:594-627, common/policies.py:752-767,
//      common/distributions.py:143-171,274-288, torch.optim.Adam, clip_grad_norm_  (same contract as ppo_train.hip).
//
// Same launch shape and exchange protocol as ppo_train.hip (3 persistent workgroups = pi | vf | cvf, weights in LDS, Adam
// moments and gradients in registers, 8-byte granules for the global gradient norm) but a different decomposition inside the
// workgroup, built to remove its barriers — the update is a chain of ~10^5 dependent optimiser steps, so the step LATENCY is
// the whole cost and every s_barrier with eight skewed waves was ~1 us of it:
//
//   * 4 waves (one per SIMD); wave w owns minibatch rows 16w..16w+15 through the WHOLE forward and the backward of the
//     activations.  All GEMMs are evaluated transposed, Z^T[feature][row] = W[feature][k] . H^T[k][row]: the weights are the
//     A operand (rows of the LDS master copy, one ds_read_b128 per four MFMA steps), the activations the B operand.  With K
//     enumerated as k = 16 js + 4 (lane/16) + e, the C layout of v_mfma_f32_16x16x4_f32 (lane holds features
//     16 t + 4 (lane/16) + i of row lane%16) IS the B-operand layout of the next layer: h1, h2, d out, dz2, dz1 never leave
//     registers and the forward + activation backward need NO barrier and no LDS round trip.
//   * the weight gradients need all 64 rows: each wave stores its 16 columns of h1^T, h2^T, dz1^T, dz2^T, dOut^T once, ONE
//     barrier, then wave w computes rows 16w.. of dW2 / dW1 and columns 16w.. of dWh with K = 64 rows; bias gradients are
//     row sums of the A operands it already holds.
//   * every wave publishes its own partial squared norm (12 granules per step instead of 3), so no reduction barrier sits
//     between the last MFMA and the publish; while the granules travel the workgroup stages the next minibatch
//     (double-buffered X) and its advantage statistics; 12 lanes poll.
//   * 3 barriers per optimiser step (activations stored | norm + staging visible | weights updated) instead of 14.
//
// Built with -ffp-contract=off; FMA is used only where written (fmaf / MFMA).
#include "ppo_common.h"

// static LDS array instead of `extern __shared__` (ppo_train_pairs.hip: -1.6 % there).  Measured here: at AntWall widths 30.5 us per
// step against 24.0 (scratch instructions 336 -> 424: the folded offsets let the optimiser hoist more than the register file holds),
// at HC widths 10.1-10.3 against 10.2-10.5 — off.
// granule store: workgroup scope (`sc0`: the line stays in this XCD's L2) when all workgroups of the run share an XCD, else agent
// scope (ppo_common.h: XCD placement)
#define GSTORE(ptr, val)                                                                              \
  do {                                                                                                \
    if (xcd_local) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  \
    else __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                \
  } while (0)
// A/B (measured and rejected twice: across the fabric in round 3, 24.0-25.1 vs 23.7-24.1 us per step, and with the XCD-local exchange in
// round 4, 20.3 vs 19.6): the partial gradients as 16-byte records {step, a, b, step} — two per group of four values — instead of four
// 8-byte granules; same bytes, half the memory instructions, bit-identical results
#ifndef ICRL_ROWS_REC16
#define ICRL_ROWS_REC16 0
#endif
// SPLIT exchange as RAW 16-byte stores + ONE flag per wave (round 5): a gradient group is four floats of a lane — one dwordx4 store, no
// tags — and when a wave has stored all its groups it drains its stores (s_waitcnt vmcnt(0): a store is acknowledged by the L2 it was
// written through to) and raises a flag word carrying the step; the same wave of the other half polls that flag and then fetches the
// groups with L1-bypassing loads.  A quarter of the memory instructions of the tagged granules (14 stores + 14 loads per lane and step
// instead of 55 + 55) and half their bytes; the exchange was request-rate-bound.  What it gives up is the early receive (nothing can be
// fetched before the flag).  Both halves still form own + partner, so they stay bit-identical replicas, and the sums are the granule
// path's sums: results are bit-identical to it.
#ifndef ICRL_ROWS_RAWX
#define ICRL_ROWS_RAWX 1
#endif
#ifndef ICRL_EARLY_PUBLISH
#define ICRL_EARLY_PUBLISH 1
#endif
#ifndef ICRL_EARLY_RECV
#define ICRL_EARLY_RECV 1
#endif
#ifndef ICRL_ROWS_STATIC_LDS
#define ICRL_ROWS_STATIC_LDS 0
#endif
// the first look at the other networks' norm granules issued before the staging of the next minibatch, read behind it (ppo_train_halves.hip)
#ifndef ICRL_ROWS_EARLY_POLL
#define ICRL_ROWS_EARLY_POLL 0
#endif

namespace icrl {

constexpr int TH4 = 256;   // 4 waves, one per SIMD (up to 512 VGPRs each)
constexpr int ST = 72;     // row stride of the [feature][row] matrices (64 rows + 8: conflict-free ds_read_b128)
constexpr int SA = 24;     // row stride of the per-row action block and of the transposed head weights

template <int NT1>
struct SmemR {  // offsets in floats (multiples of 4)
  static constexpr int O16 = 16 * NT1, SX = O16 + 8;
  static constexpr bool XDB = NT1 <= 2;        // second X buffer (next minibatch staged while dW1 still reads this one)
  static constexpr bool W2TC = NT1 <= 2;       // transposed copy of W2 (ds_read_b128 operand fetch in the backward); LDS budget
  static constexpr bool WHTC = NT1 <= 4;       // transposed copy of the head weights
  static constexpr bool DZ1A = NT1 > 4;        // dz1^T shares h2^T's storage (written after dW2 / dWh have read h2^T): LDS budget
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
  static constexpr int PST = ADC + RB;         // [4][8] per-wave loss statistics
  static constexpr int PLS = PST + 32;         // [4][16] per-wave d log_std partial sums
  static constexpr int MISC = PLS + 64;        // [64] granule values, flags, advantage-statistics partials
  static constexpr int TOTAL = MISC + 64;
};

// the kernel arguments, re-read from the kernel-argument segment (scalar loads, scalar-cache resident) at the few places
// that need pointers: keeps ~30 scalar registers out of the loop-carried state (the SGPR file spills otherwise)
#define KARGS() ([&]() { const TrainArgs* k_ = ka; asm volatile("" : "+s"(k_)); return k_; }())

// SPLIT: TWO workgroups per network (grid 6).  A minibatch of more than 64 rows is two chunks; workgroup (role, half) runs
// the forward / backward / weight-gradient GEMMs of chunk `half` only, the two halves then exchange their partial gradients
// (every gradient register as a {step tag | float} granule, same protocol as the norm exchange) and each forms
// own + partner — float addition is commutative bit for bit, so both halves hold identical gradients, run the identical norm /
// Adam arithmetic on identical weights and stay replicas of each other.  One more hop per step for half of the GEMM work.
// the whole update of ONE run (see ppo_train_pairs_body): `ka` = the same argument block in memory
// BATCH: the argument block was read from memory (batched launch): its pointers are marked as global-memory pointers (common.h:
// as_global).  The single-run kernel's by-value pointers already are, and at AntWall widths the extra scalar traffic costs it 6 %.
// SPLIT exchange: which gradient group (0 .. NT1-1: W1 tiles, NT1 .. NT1+3: W2 tiles, NT1+4: head, NT1+5: {b1, b2, extra}) is fetched
// during dW1 tile c — the W2 tiles and the head first (published before dW1 started), then the W1 tiles published five tiles ago
__host__ __device__ constexpr int early_group_of(int nt1, int c) { return c < 5 ? nt1 + c : c - 5; }
template <int NT1>
struct RemGroups {      // the groups the final pass still has to fetch
  int v[NT1 + 6];
  int n;
  static constexpr RemGroups make(bool early) {
    RemGroups r{};
    r.n = 0;
    for (int g = 0; g < NT1 + 6; ++g) {
      bool e = false;
      for (int c = 0; c < NT1; ++c) e = e || (early && early_group_of(NT1, c) == g);
      if (!e) r.v[r.n++] = g;
    }
    return r;
  }
};

template <int NT1, bool DISC, bool SPLIT, bool BATCH>
__device__ __forceinline__ void ppo_train_rows_body(const TrainArgs& a, const TrainArgs* const ka, const int slot_j) {
#define GP(x) (BATCH ? as_global(x) : (x))
  using S = SmemR<NT1>;
  constexpr int SX = S::SX;
  constexpr int XR = (S::O16 + 3) / 4;  // floats of an X row each of the 4 threads of a row stages
#if ICRL_ROWS_STATIC_LDS
  __shared__ __attribute__((aligned(16))) float sm[S::TOTAL];   // static: offsets fold into immediates (see ppo_train_pairs.hip)
#else
  extern __shared__ __attribute__((aligned(16))) float sm[];
#endif
  const int role = slot_j % 3;   // 0 policy, 1 reward critic, 2 cost critic
  const int half = slot_j / 3;   // SPLIT: which 64-row chunk of every minibatch this workgroup computes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int O = a.L.O, A = a.L.A;
  const int n_out = role == 0 ? A : 1;
  const int T = a.buf.T, N = a.buf.N;
  const float nu = GP(a.nu)[0];
  const int n_steps = a.n_steps;
  const PlanStep* __restrict__ const plan_steps = GP(a.plan_steps);
  const PlanChunk* __restrict__ const plan_chunks = GP(a.plan_chunks);
  const int* __restrict__ const perms = GP(a.perms);
  // per-row side data of this role, selected once:
  //   policy: old log-prob, reward advantage, cost advantage;  critics: old value, return, (unused)
  const float* const p_s0 = GP(role == 0 ? a.buf.log_probs : (role == 1 ? a.buf.reward_values : a.buf.cost_values));
  const float* const p_s1 = GP(role == 0 ? a.buf.reward_advantages : (role == 1 ? a.buf.reward_returns : a.buf.cost_returns));
  const float* const p_s2 = GP(a.buf.cost_advantages);
  const float* const p_obs = GP(a.buf.observations);
  const float* const p_act = GP(a.buf.actions);
  const int AS = a.buf.act_store;

  // wave w owns W1 / W2 rows 16w..16w+15 (element j = 16w + 4q + i, k = 16c + r) and head-weight columns 16w..16w+15
  // (element o = 4q + i, j = 16w + r); b1 / b2 entries 16w + r (replicated over q, lane q == 0 stores); wave 0: head bias r,
  // wave 1: log_std r.  The owner keeps the fp32 master value, both Adam moments and the accumulating gradient in registers;
  // LDS holds the operand copies every wave reads.
  // Register plan (one wave per SIMD): the fp32 MASTER weights live in LDS only (the operand copies ARE the master values; the
  // owner lane re-reads its elements for the Adam step), both Adam moments live in the accumulation-register half of the
  // register file and are touched only inside the Adam step (explicit v_accvgpr moves: 2 reads + 2 writes per element and
  // step), the gradient accumulators are MFMA destinations.  This keeps ~90 long-lived values out of the 256 VALU-addressable
  // registers, which the forward / backward phases need for operand pipelining.
  constexpr int E_W2 = 4 * NT1, E_WH = E_W2 + 16, E_B1 = E_WH + 4, E_B2 = E_B1 + 1, E_EX = E_B2 + 1, NEL = E_EX + 1;
  float mA[NEL], vA[NEL];                      // AGPR-resident (acc_put / acc_get)
  f32x4 gW1r[NT1], gW2r[4], gWhr;
  const int jb = 16 * w + r;
  float gb1r = 0.f, gb2r = 0.f, gex = 0.f;
  int ex_g = -1, ex_s = S::MISC + 63;          // "extra" vector entry: wave 0 head bias r, wave 1 log_std r (Gaussian policy)
  {
    const PolLayout& L = a.L;
    const int gbh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);
    if (w == 0 && r < n_out) { ex_g = gbh + r; ex_s = S::BH + r; }
    if (w == 1 && !DISC && role == 0 && r < A) { ex_g = L.log_std + r; ex_s = S::LS + r; }
  }
  // operand copies (= master values): every owner lane stores / re-reads its own elements
  auto store_w1 = [&](int c, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[S::W1 + (16 * w + 4 * q + i) * SX + 16 * c + r] = v[i];
  };
  auto load_own_w1 = [&](int c) -> f32x4 {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[S::W1 + (16 * w + 4 * q + i) * SX + 16 * c + r];
    return v;
  };
  auto store_w2 = [&](int c, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[S::W2 + (16 * w + 4 * q + i) * SH + 16 * c + r] = v[i];
    if (S::W2TC) *reinterpret_cast<f32x4*>(sm + S::W2T + (16 * c + r) * SH + 16 * w + 4 * q) = v;
  };
  auto load_own_w2 = [&](int c) -> f32x4 {
    if (S::W2TC) return lds128(sm + S::W2T + (16 * c + r) * SH + 16 * w + 4 * q);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[S::W2 + (16 * w + 4 * q + i) * SH + 16 * c + r];
    return v;
  };
  auto store_wh = [&](const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[S::WH + (4 * q + i) * SH + 16 * w + r] = v[i];
    if (S::WHTC) *reinterpret_cast<f32x4*>(sm + S::WHT + (16 * w + r) * SA + 4 * q) = v;
  };
  auto load_own_wh = [&]() -> f32x4 {
    if (S::WHTC) return lds128(sm + S::WHT + (16 * w + r) * SA + 4 * q);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[S::WH + (4 * q + i) * SH + 16 * w + r];
    return v;
  };
  for (int i = tid; i < S::TOTAL; i += TH4) sm[i] = 0.f;
  __syncthreads();
  {
    const PolLayout& L = a.L;
    const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
    const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
#pragma unroll
    for (int c = 0; c < NT1; ++c) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * w + 4 * q + i, k = 16 * c + r;
        pv[i] = k < O ? a.params[gW1 + j * O + k] : 0.f;
        acc_put(mA[4 * c + i], k < O ? a.exp_avg[gW1 + j * O + k] : 0.f);
        acc_put(vA[4 * c + i], k < O ? a.exp_avg_sq[gW1 + j * O + k] : 0.f);
      }
      store_w1(c, pv);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * w + 4 * q + i, k = 16 * c + r;
        pv[i] = a.params[gW2 + j * HD + k];
        acc_put(mA[E_W2 + 4 * c + i], a.exp_avg[gW2 + j * HD + k]);
        acc_put(vA[E_W2 + 4 * c + i], a.exp_avg_sq[gW2 + j * HD + k]);
      }
      store_w2(c, pv);
    }
    {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = 4 * q + i, j = 16 * w + r;
        pv[i] = o < n_out ? a.params[gWh + o * HD + j] : 0.f;
        acc_put(mA[E_WH + i], o < n_out ? a.exp_avg[gWh + o * HD + j] : 0.f);
        acc_put(vA[E_WH + i], o < n_out ? a.exp_avg_sq[gWh + o * HD + j] : 0.f);
      }
      store_wh(pv);
    }
    acc_put(mA[E_B1], a.exp_avg[gb1 + jb]); acc_put(vA[E_B1], a.exp_avg_sq[gb1 + jb]);
    acc_put(mA[E_B2], a.exp_avg[gb2 + jb]); acc_put(vA[E_B2], a.exp_avg_sq[gb2 + jb]);
    acc_put(mA[E_EX], ex_g >= 0 ? a.exp_avg[ex_g] : 0.f); acc_put(vA[E_EX], ex_g >= 0 ? a.exp_avg_sq[ex_g] : 0.f);
    if (q == 0) {
      sm[S::B1 + jb] = a.params[gb1 + jb]; sm[S::B2 + jb] = a.params[gb2 + jb];
      sm[ex_s] = ex_g >= 0 ? a.params[ex_g] : 0.f;          // lanes without an extra entry hit a scratch word
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
  u64* const xch = GP(a.xch);

  // ---------------------------------------------------------------------------------------------------------------
  // row stream: rows of chunk g+1 are prefetched into registers while chunk g is processed, their permutation indices two
  // chunks earlier; the schedule comes from the plan tables (uniform scalar loads)
  // ---------------------------------------------------------------------------------------------------------------
  // `perms` holds STORAGE offsets here: ppo_perm_offsets_kernel has already mapped the flat env-major indices of the
  // permutations (ref: buffers.py:53-65, i -> env = i / T, t = i % T) to t * N + env

  // this thread's row of a chunk: b = tid/4 — a row of wave w's own tile, so the per-row side data is wave-private
  const int gb_row = tid >> 2, gpart = tid & 3;
  // Both index streams are software-pipelined one stage deeper than the data they address: a plan entry is loaded one
  // step / chunk before the permutation entry it locates (a dependent pair of global loads would otherwise stall the wave
  // for a full memory round trip at the point where the second address is formed).
  const int stid = tid - 64;      // advantage statistics: row stid of the minibatch lives in waves 1 (and 2); wave 0 polls
  // Plan entries are fetched as VECTOR values and kept opaque until the step / chunk they describe becomes the current one:
  // left to itself the compiler scalarises a uniform load at once (v_readfirstlane right behind the load = a full memory
  // round trip per step, plus the spill traffic of three rotating 4-dword structs in the scalar file).
  // (the INDEX is laundered through a vector register, not the loaded value: an asm operand on the value would itself wait
  // for the load)
  auto ld_step = [&](int i) -> int4 {
    asm volatile("" : "+v"(i));
    return *reinterpret_cast<const int4*>(plan_steps + i);
  };
  auto ld_chunk = [&](int g) -> int2 {
    asm volatile("" : "+v"(g));
    return *reinterpret_cast<const int2*>(plan_chunks + (SPLIT ? 2 * g + half : g));   // SPLIT: the plan holds two entries per step
  };
  auto chunk_idx = [&](const int2& c) -> int { return gb_row < c.y ? perms[c.x + gb_row] : -1; };      // {perm_base, rows}
  auto stat_idx = [&](const int4& p) -> int {       // row stid of that step's minibatch (policy role); p.z = nb_flags, p.w = perm_base
    return (role == 0 && stid >= 0 && stid < (p.z & NB_MASK)) ? perms[p.w + stid] : -1;
  };
  // minibatches of 193..256 rows (one workgroup walks four chunks): the 192 statistics threads take a second row, stid + 192
  const bool big_mb = !SPLIT && a.hp.batch_size > 192;
  auto stat_idx2 = [&](const int4& p) -> int {
    return (role == 0 && stid >= 0 && stid + 192 < (p.z & NB_MASK)) ? perms[p.w + stid + 192] : -1;
  };
  float px[XR], pact[4], psc0 = 0.f, psc1 = 0.f, psc2 = 0.f;
  bool pvalid = false;
  auto issue_rows = [&](int idx) {
    // every load is issued unconditionally at a clamped (always valid) address; the values are masked only when they are
    // committed to LDS, so nothing waits on the loads here and all of them are in flight together
    pvalid = idx >= 0;
    const size_t off = pvalid ? (size_t)idx : 0;
    const float* orow = p_obs + off * O;
    // component k = gpart + 4 i: the four threads of a row read 16 consecutive bytes per instruction
#pragma unroll
    for (int i = 0; i < XR; ++i) { const int k = gpart + 4 * i; px[i] = orow[k < O ? k : O - 1]; }
    if (role == 0) {
      const float* arow = p_act + off * AS;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int k = gpart + 4 * i; pact[i] = arow[k < AS ? k : AS - 1]; }
    }
    psc0 = p_s0[off]; psc1 = p_s1[off]; psc2 = p_s2[off];
  };
  auto commit_rows = [&](int xbase) {
#pragma unroll
    for (int i = 0; i < XR; ++i) { const int k = gpart + 4 * i; sm[xbase + k * ST + gb_row] = (pvalid && k < O) ? px[i] : 0.f; }
    if (role == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int k = gpart + 4 * i; sm[S::ACT + gb_row * SA + k] = (pvalid && k < AS) ? pact[i] : 0.f; }
    }
    if (gpart == 0) { sm[S::OLP + gb_row] = pvalid ? psc0 : 0.f; sm[S::ADR + gb_row] = pvalid ? psc1 : 0.f; sm[S::ADC + gb_row] = pvalid ? psc2 : 0.f; }
  };
  // advantage statistics of a minibatch (policy role): thread 64 + i, i < nb (<= 128: waves 1 and 2) holds row i's (A_r, A_c)
  float sar = 0.f, sac = 0.f, sar2 = 0.f, sac2 = 0.f;
  auto issue_stats = [&](int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;     // rows beyond the minibatch are masked in stats_partials
    sar = p_s1[off];                                       // (only the policy role uses them)
    sac = p_s2[off];
  };
  auto issue_stats2 = [&](int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;
    sar2 = p_s1[off];
    sac2 = p_s2[off];
  };
  // per-wave partial sums into MISC[32..40] (waves 1, 2; wave 3 as well when a minibatch has more than 128 rows); combined by
  // everybody after the next barrier (fixed order)
  const bool wide_mb = !SPLIT && a.hp.batch_size > 128;
  auto stats_partials = [&](int nb) {
    if (role != 0 || w == 0 || (w == 3 && !wide_mb)) return;
    const bool in = stid < nb, in2 = big_mb && stid + 192 < nb;
    const float v_r = (in ? sar : 0.f) + (in2 ? sar2 : 0.f), v_c = (in ? sac : 0.f) + (in2 ? sac2 : 0.f);
    const float v_rr = (in ? sar * sar : 0.f) + (in2 ? sar2 * sar2 : 0.f);
    const float s_r = wave_sum_fast(v_r), s_c = wave_sum_fast(v_c), s_rr = wave_sum_fast(v_rr);
    if (lane == 0) { sm[S::MISC + 29 + 3 * w] = s_r; sm[S::MISC + 30 + 3 * w] = s_c; sm[S::MISC + 31 + 3 * w] = s_rr; }
  };
  float mean_r = 0.f, istd_r = 1.f, mean_c = 0.f;
  auto read_stats = [&](int nb) {
    if (role != 0) return;
    float s_r = sm[S::MISC + 32] + sm[S::MISC + 35];
    float s_c = sm[S::MISC + 33] + sm[S::MISC + 36];
    float s_rr = sm[S::MISC + 34] + sm[S::MISC + 37];
    if (wide_mb) { s_r += sm[S::MISC + 38]; s_c += sm[S::MISC + 39]; s_rr += sm[S::MISC + 40]; }
    const float inv = __builtin_amdgcn_rcpf((float)nb);
    mean_r = s_r * inv;
    mean_c = s_c * inv;
    // unbiased variance from the raw moments (advantages are O(1): fp32 cancellation stays ~1e-6 relative)
    const float var = fmaxf(s_rr - s_r * mean_r, 0.f) * __builtin_amdgcn_rcpf((float)(nb - 1));
    istd_r = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) + 1e-8f);
  };
  auto refresh_gauss = [&]() {   // wave 1, lanes q == 0 own log_std r: derived constants of the Gaussian head
    if (!DISC && role == 0 && w == 1 && q == 0) {
      const float wex = sm[S::LS + r];
      const float sd = __expf(wex);
      const float iv = __builtin_amdgcn_rcpf(sd * sd);
      sm[S::GAU + r] = r < A ? iv : 0.f;
      sm[S::GAU + 16 + r] = r < A ? 0.5f * iv : 0.f;
      sm[S::GAU + 32 + r] = r < A ? wex + LOG_SQRT_2PI_F : 0.f;     // log(sd) = log_std
      // entropy of the diagonal Gaussian, sum_a 0.5 + 0.5 log(2 pi) + log sigma_a (one value per step for the statistics)
      const float ent = row_sum(r < A ? HALF_LOG_2PI_PLUS_HALF_F + wex : 0.f);
      if (r == 0) sm[S::MISC + 22] = ent;
    }
  };
  // the running statistics live in lane 0 of wave 3 of each role: wave 0 polls the granules, the other waves would idle at
  // the barrier anyway, so the bookkeeping stays off the path that decides when the workgroup can go on
  const bool book = tid == 192;
  float st_ent = 0.f, st_pg = 0.f, st_vl = 0.f, st_cf = 0.f, last_loss = 0.f, kl_sum = 0.f;
  int steps_done = 0, early_stop_epoch = a.hp.n_epochs, status = 0;

  // ---- pipeline prologue
  int g_chunk = 0;                      // global index of the chunk being processed
  int idx_next = chunk_idx(ld_chunk(1)), idx_nx2 = chunk_idx(ld_chunk(2));
  int2 pc_nx3 = ld_chunk(3);            // entry of chunk g + 3, loaded one chunk before its indices are
  issue_rows(chunk_idx(ld_chunk(0)));
  int4 ps_next = ld_step(0), ps_nx2 = ld_step(1), ps_nx3 = ld_step(2);   // steps st, st + 1, st + 2 at the loop top
  issue_stats(stat_idx(ps_next));
  int sidx_next = stat_idx(ps_nx2);
  int sidx2_next = -1;
  if (big_mb) { issue_stats2(stat_idx2(ps_next)); sidx2_next = stat_idx2(ps_nx2); }
  refresh_gauss();
  int xcur = S::XT0;                    // X^T buffer of the chunk being processed
  commit_rows(xcur);
  stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  if (tid == 0) sm[S::MISC + 14] = run_on_one_xcd(xch, slot_j, SPLIT ? 6 : 3) ? 1.f : 0.f;
  __syncthreads();
  const bool xcd_local = __builtin_amdgcn_readfirstlane(__float_as_int(sm[S::MISC + 14])) != 0;
  read_stats(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  const float inv_n_mb = 1.f / (float)((T * N + a.hp.batch_size - 1) / a.hp.batch_size);

  const bool prof = (a.hp._pad & 1) != 0;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = prof ? stamp() : 0ull;

  bool stop = false;
  for (int st = 0; st < n_steps && !stop; ++st) {
    const unsigned step = (unsigned)st + 1u;
    PlanStep ps;                           // this step's entry becomes scalar only now (it was loaded three steps ago)
    ps.step_size = __int_as_float(__builtin_amdgcn_readfirstlane(ps_next.x));
    ps.inv_bc2_sqrt = __int_as_float(__builtin_amdgcn_readfirstlane(ps_next.y));
    ps.nb_flags = __builtin_amdgcn_readfirstlane(ps_next.z);
    ps.perm_base = 0;
    ps_next = ps_nx2; ps_nx2 = ps_nx3;
    ps_nx3 = ld_step(st + 3 < n_steps + 2 ? st + 3 : n_steps + 1);     // (the table carries two zero entries at its end)
    const int nb = ps.nb_flags & NB_MASK;
    const float inv_nb = __builtin_amdgcn_rcpf((float)nb);
    const float c_mean_r = mean_r, c_mean_c = mean_c, c_istd_r = istd_r;   // statistics of THIS minibatch
    const float cpol_nb = inv_nb * __builtin_amdgcn_rcpf(1.f + nu);
    issue_stats(sidx_next);              // advantages of the NEXT minibatch's rows (indices loaded a step ago) ...
    sidx_next = stat_idx(ps_nx2);        // ... and the indices of the one after (step st + 2; its plan entry came a step ago)
    if (big_mb) { issue_stats2(sidx2_next); sidx2_next = stat_idx2(ps_nx2); }
    float mb_s0 = 0.f, mb_s1 = 0.f, mb_s2 = 0.f, mb_s3 = 0.f, mb_s4 = 0.f;  // bookkeeping lane: minibatch sums of the loss statistics

    const int n_chunks = SPLIT ? 1 : (nb + RB - 1) / RB;
    unsigned pend = 0; (void)pend;      // SPLIT: gradient groups of the other half that had not arrived when they were looked at
#if ICRL_ROWS_RAWX
    typedef unsigned int raw_u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(SPLIT ? GP(KARGS()->gx) : nullptr, 0, (int)ICRL_PPO_SPLIT_BYTES, 0x00020000);
    constexpr int XBLK = (4 * NT1 + 23 + 5) * TH4 * 8;          // bytes of one (parity, role, half) block (the granule layout's size)
    constexpr int XFLAG = (NT1 + 7) * TH4 * 16;                 // NT1 + 6 groups + the book-keeping lane's record, then one flag word per wave (64 B apart)
    static_assert(XFLAG + 4 * 64 <= XBLK, "raw exchange block");
    const int xmine = (((int)(step & 1) * 3 + role) * 2 + half) * XBLK, xtheirs = (((int)(step & 1) * 3 + role) * 2 + (1 - half)) * XBLK;
    auto raw_store = [&](int byte_off, const f32x4& v) {
      const raw_u4 u = __builtin_bit_cast(raw_u4, v);
      if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(u, grs, byte_off, 0, 1);      // sc0: stays in this XCD's L2
      else __builtin_amdgcn_raw_buffer_store_b128(u, grs, byte_off, 0, 16);               // sc1
    };
    auto raw_load = [&](int byte_off) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, byte_off, 0, 16)); };
#elif ICRL_ROWS_REC16
    typedef unsigned int rec_u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(SPLIT ? GP(KARGS()->gx) : nullptr, 0, (int)ICRL_PPO_SPLIT_BYTES, 0x00020000);
    constexpr int RBLK = (4 * NT1 + 23 + 5) * TH4 * 8;          // bytes of one (parity, role, half) block: (KG + 5) / 2 records per thread
    const int rmine = (((int)(step & 1) * 3 + role) * 2 + half) * RBLK + tid * 16;
    const int rtheirs = (((int)(step & 1) * 3 + role) * 2 + (1 - half)) * RBLK + tid * 16;
    auto rec_store = [&](int byte_off, rec_u4 v) {
      if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(v, grs, byte_off, 0, 1);      // sc0
      else __builtin_amdgcn_raw_buffer_store_b128(v, grs, byte_off, 0, 16);               // sc1
    };
    auto rec_load = [&](int byte_off) -> rec_u4 { return __builtin_amdgcn_raw_buffer_load_b128(grs, byte_off, 0, 16); };
    auto rec_pub = [&](int g, const f32x4& v) {      // group g = records 2 g, 2 g + 1 of this thread
      rec_store(rmine + (2 * g) * TH4 * 16, rec_u4{step, __float_as_uint(v[0]), __float_as_uint(v[1]), step});
      rec_store(rmine + (2 * g + 1) * TH4 * 16, rec_u4{step, __float_as_uint(v[2]), __float_as_uint(v[3]), step});
    };
    auto rec_ok = [&](const rec_u4& a_, const rec_u4& b_) -> bool { return a_[0] == step && a_[3] == step && b_[0] == step && b_[3] == step; };
#endif
    for (int ch = 0; ch < n_chunks; ++ch, ++g_chunk) {
      const int first = SPLIT ? half * RB : ch * RB;        // SPLIT: a half without rows (ragged last minibatch) runs on zero rows
      const int nrows = nb - first < 0 ? 0 : ((nb - first) < RB ? (nb - first) : RB);
      if (ch > 0) {  // chunk 0 of a step was committed during the previous step's granule wait (or the prologue); the
        // chunk-end barrier below separates this from the previous chunk's readers, and the forward reads only this wave's
        // own rows, which this wave's threads stage
        if (S::XDB) xcur = xcur == S::XT0 ? S::XT1 : S::XT0;
        commit_rows(xcur);
      }
      const int b = 16 * w + r;             // this lane's row of the chunk (all four q lanes share it)
      const bool valid = b < nrows;
      // ================= forward, transposed: rows of the weights x this wave's 16 rows =================
      // Operand fetches are software-pipelined by hand: the ds_reads of tile t + 1 (or of the next layer's first tile) are
      // issued before the MFMAs of tile t, with a scheduling fence so that they stay there — a wave otherwise stalls for
      // the full LDS latency (~130 cycles) in front of every group of MFMAs, ~25 times per step.
      f32x4 h1c[4], h2c[4], outc;
      f32x4 w2p[2][4], b2p[2];              // W2 tile operands / bias, ping-pong
      auto load_w2 = [&](int t, f32x4 (&aw)[4], f32x4& bias) {
        const float* pa = sm + S::W2 + (16 * t + r) * SH + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) aw[js] = lds128(pa + 16 * js);
        bias = lds128(sm + S::B2 + 16 * t + 4 * q);
      };
      {
        float bx[NT1][4];                   // x[row b][k = 16 js + 4q + e]
        const float* pb = sm + xcur + (4 * q) * ST + b;
#pragma unroll
        for (int js = 0; js < NT1; ++js)
#pragma unroll
          for (int e = 0; e < 4; ++e) bx[js][e] = pb[(16 * js + e) * ST];
        f32x4 w1p[2][NT1], b1p[2];
        auto load_w1 = [&](int t, f32x4 (&aw)[NT1], f32x4& bias) {
          const float* pa = sm + S::W1 + (16 * t + r) * SX + 4 * q;
#pragma unroll
          for (int js = 0; js < NT1; ++js) aw[js] = lds128(pa + 16 * js);
          bias = lds128(sm + S::B1 + 16 * t + 4 * q);
        };
        constexpr bool PF1 = NT1 <= 4;      // wide observations: a second W1 operand tile in flight would spill registers
        load_w1(0, w1p[0], b1p[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < 3) { if (PF1) load_w1(t + 1, w1p[(t + 1) & 1], b1p[(t + 1) & 1]); }
          else load_w2(0, w2p[0], b2p[0]);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc = b1p[t & 1];             // the bias is the accumulator's initial value (C input of the first MFMA)
#pragma unroll
          for (int js = 0; js < NT1; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (js < NT1 - 1 || 16 * js + e < O)      // last k-block: components 16 js + 4 q + e all beyond obs -> nothing to add
                acc = MFMA_F32(w1p[t & 1][js][e], bx[js][e], acc);
#pragma unroll
          for (int i = 0; i < 4; ++i) h1c[t][i] = fast_tanh(acc[i]);
          if (!PF1 && t < 3) load_w1(t + 1, w1p[(t + 1) & 1], b1p[(t + 1) & 1]);
        }
      }
      // prefetch: rows of the next chunk of the stream, indices of the chunk three ahead.  Issued this early on purpose: the
      // gathered rows are random 72-byte pieces of a 47 MB buffer and take several microseconds to arrive (measured: issuing
      // them after the activation backward instead costs 4 000 cycles per step)
      issue_rows(idx_next);
      idx_next = idx_nx2;
      idx_nx2 = chunk_idx(pc_nx3);
      pc_nx3 = ld_chunk(g_chunk + 4);
      f32x4 whp[4], bhp;                    // head operands
      {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < 3) load_w2(t + 1, w2p[(t + 1) & 1], b2p[(t + 1) & 1]);
          else {
            const float* pa = sm + S::WH + r * SH + 4 * q;
#pragma unroll
            for (int js = 0; js < 4; ++js) whp[js] = lds128(pa + 16 * js);
            bhp = lds128(sm + S::BH + 4 * q);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc = b2p[t & 1];
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = MFMA_F32(w2p[t & 1][js][e], h1c[js][e], acc);
#pragma unroll
          for (int i = 0; i < 4; ++i) h2c[t][i] = fast_tanh(acc[i]);
        }
      }
      {  // head: outputs o = 4q + i of row b; four independent chains
        f32x4 acc[4];
#pragma unroll
        for (int js = 0; js < 4; ++js) {
          acc[js] = js == 0 ? bhp : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[js] = MFMA_F32(whp[js][e], h2c[js][e], acc[js]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) outc[i] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
      }
      // this wave's 16 columns of h1^T / h2^T (read by the weight-gradient GEMMs after the barrier)
      float* const pt = sm + (4 * q) * ST + b;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pt[S::H1T + (16 * t + i) * ST] = h1c[t][i];
          pt[S::H2T + (16 * t + i) * ST] = h2c[t][i];
        }
      STAMP(0)   // forward
      // ============ loss + d loss / d head output, in the C layout (== B operand of the backward) ============
      f32x4 dout = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;   // per-row statistics (identical in the four q lanes)
        if (role == 0) {
          float lp = 0.f, ent = 0.f;
          f32x4 g1 = f32x4{0.f, 0.f, 0.f, 0.f}, g2 = f32x4{0.f, 0.f, 0.f, 0.f};   // d log-prob / d out, second term
          if (DISC) {
            // Categorical(logits) (ref: distributions.py:274-288)
            float lg[4], zmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) { lg[i] = (4 * q + i < A) ? outc[i] : -INFINITY; zmax = fmaxf(zmax, lg[i]); }
            zmax = xor16_max(xor32_max(zmax));
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) se += (4 * q + i < A) ? expf(lg[i] - zmax) : 0.f;
            se = quad_rows_sum(se);
            const float lse = zmax + logf(se);
            const int act = (int)sm[S::ACT + b * SA];
            float pr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = 4 * q + i;
              lg[i] = k < A ? lg[i] - lse : 0.f;
              pr[i] = k < A ? expf(lg[i]) : 0.f;
              lp += (k == act) ? lg[i] : 0.f;
              ent -= pr[i] * lg[i];
            }
            lp = quad_rows_sum(lp);
            ent = quad_rows_sum(ent);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = 4 * q + i;
              g1[i] = k < A ? ((k == act ? 1.f : 0.f) - pr[i]) : 0.f;
              g2[i] = k < A ? pr[i] * (lg[i] + ent) : 0.f;     // d(-H)/dz_k = p_k (log p_k + H)
            }
          } else {
            const f32x4 actv = lds128(sm + S::ACT + b * SA + 4 * q);
            const f32x4 iv = lds128(sm + S::GAU + 4 * q), hiv = lds128(sm + S::GAU + 16 + 4 * q), lsd = lds128(sm + S::GAU + 32 + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float dd = actv[i] - outc[i];                 // pad actions / outputs are 0
              lp += -(dd * dd) * hiv[i] - lsd[i];
              g1[i] = dd * iv[i];
              g2[i] = (4 * q + i < A) ? (dd * dd) * iv[i] - 1.f : 0.f;   // d log-prob / d log_std
            }
            lp = quad_rows_sum(lp);
          }
          const float old_lp = sm[S::OLP + b];
          const float ratio = __expf(lp - old_lp);
          const float Ar = (sm[S::ADR + b] - c_mean_r) * c_istd_r;
          const float Ac = sm[S::ADC + b] - c_mean_c;
          const float s1 = Ar * ratio;
          const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
          const float s2 = Ar * rc;
          const float gsel = (s1 <= s2) ? Ar : 0.f;                       // d min(s1, s2) / d ratio
          const float dlp = valid ? cpol_nb * (-gsel + nu * Ac) * ratio : 0.f;  // d loss / d log_prob
          if (DISC) {
            const float dent = valid ? ent_coef * inv_nb : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) dout[i] = dlp * g1[i] + dent * g2[i];
          } else {
            f32x4 t;       // d log_std: sum over this wave's 16 rows, per output o = 4q + i
#pragma unroll
            for (int i = 0; i < 4; ++i) { dout[i] = dlp * g1[i]; t[i] = row_sum(dlp * g2[i]); }
            if (r == 0) *reinterpret_cast<f32x4*>(sm + S::PLS + 16 * w + 4 * q) = t;
          }
          const bool cnt = valid && q == 0;
          v0 = cnt ? fminf(s1, s2) : 0.f; v1 = cnt ? Ac * ratio : 0.f; v2 = (cnt && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
          v3 = cnt ? old_lp - lp : 0.f; v4 = cnt ? ent : 0.f;
        } else {
          const float v = quad_rows_sum(q == 0 ? outc[0] : 0.f);      // lane (r, q = 0) holds output 0 of row b
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
        // only the q == 0 lanes carry a value: sum over the 16 lanes of row 0
        v0 = row_sum(v0); v1 = row_sum(v1); v2 = row_sum(v2); v3 = row_sum(v3);
        if (DISC) v4 = row_sum(v4);
        if (lane == 0) { float* pst = sm + S::PST + 8 * w; pst[0] = v0; pst[1] = v1; pst[2] = v2; pst[3] = v3; pst[4] = v4; }
      }
      STAMP(1)   // loss
      // ================= backward of the activations (registers only) =================
      f32x4 dz2c[4], dz1c[4];
      {  // dH2^T = Wh^T . dOut^T: A = WHT[j = 16t + r][o = 4q + e] (K = 16 outputs)
        const float* pa = S::WHTC ? sm + S::WHT + r * SA + 4 * q : sm + S::WH + (4 * q) * SH + r;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 aw;
          if (S::WHTC) aw = lds128(pa + t * 16 * SA);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) aw[e] = pa[e * SH + 16 * t];
          }
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = MFMA_F32(aw[e], dout[e], acc);
#pragma unroll
          for (int i = 0; i < 4; ++i) dz2c[t][i] = fmaf(-(h2c[t][i] * h2c[t][i]), acc[i], acc[i]);   // acc (1 - h2^2)
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::DZ2T + (16 * t + i) * ST] = dz2c[t][i];
#pragma unroll
      for (int i = 0; i < 4; ++i) pt[S::DOT + i * ST] = dout[i];
      {  // dH1^T = W2^T . dz2^T: A = W2T[k = 16t + r][j = 16 js + 4q + e]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 aw[4];
          if (S::W2TC) {
            const float* pa = sm + S::W2T + (16 * t + r) * SH + 4 * q;
#pragma unroll
            for (int js = 0; js < 4; ++js) aw[js] = lds128(pa + 16 * js);
          } else {
            const float* pa = sm + S::W2 + (4 * q) * SH + 16 * t + r;
#pragma unroll
            for (int js = 0; js < 4; ++js)
#pragma unroll
              for (int e = 0; e < 4; ++e) aw[js][e] = pa[(16 * js + e) * SH];
          }
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = MFMA_F32(aw[js][e], dz2c[js][e], acc);
#pragma unroll
          for (int i = 0; i < 4; ++i) dz1c[t][i] = fmaf(-(h1c[t][i] * h1c[t][i]), acc[i], acc[i]);
        }
      }
      if (!S::DZ1A) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::DZ1T + (16 * t + i) * ST] = dz1c[t][i];
      }
      lds_barrier();  // (B1) all 64 rows of the activations / their gradients are visible
      STAMP(2)   // activation backward
      if (ch == 0) {   // gradient accumulators start their life here
#pragma unroll
        for (int c = 0; c < NT1; ++c) gW1r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) gW2r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        gWhr = f32x4{0.f, 0.f, 0.f, 0.f};
        gb1r = 0.f; gb2r = 0.f; gex = 0.f;
      }
      // SPLIT + ICRL_EARLY_PUBLISH: a gradient register group goes out to the other half as soon as its GEMM is done — one GEMM late, so that
      // the store does not wait for the MFMA chain that produces it — instead of all 55 granules per thread behind the last GEMM:
      // the exchange is throughput-bound (~1 granule per cycle and CU), and the memory pipe runs beside the MFMAs of the next tiles.
      constexpr int KG_ = 4 * NT1 + 23;
#if !ICRL_ROWS_RAWX
      u64* const mine_e = SPLIT ? GP(KARGS()->gx) + ((size_t)((step & 1) * 3 + role) * 2 + half) * ((size_t)(KG_ + 5) * TH4) + tid : nullptr;
#endif
      (void)KG_;
#if ICRL_ROWS_RAWX
      auto publish4 = [&](int g, const f32x4& v) {
        if (!(SPLIT && ICRL_EARLY_PUBLISH)) return;
        raw_store(xmine + (g * TH4 + tid) * 16, v);
      };
      auto early_issue = [&](int) {};
      auto early_take = [&](int, f32x4&) {};
#elif ICRL_ROWS_REC16
      auto publish4 = [&](int g, const f32x4& v) {
        if (!(SPLIT && ICRL_EARLY_PUBLISH)) return;
        rec_pub(g, v);
      };
      rec_u4 eb[2] = {rec_u4{0, 0, 0, 0}, rec_u4{0, 0, 0, 0}};
      auto early_issue = [&](int g) { eb[0] = rec_load(rtheirs + (2 * g) * TH4 * 16); eb[1] = rec_load(rtheirs + (2 * g + 1) * TH4 * 16); };
      auto early_take = [&](int g, f32x4& v) {
        const bool ok = rec_ok(eb[0], eb[1]);
        v[0] = ok ? v[0] + __uint_as_float(eb[0][1]) : v[0]; v[1] = ok ? v[1] + __uint_as_float(eb[0][2]) : v[1];
        v[2] = ok ? v[2] + __uint_as_float(eb[1][1]) : v[2]; v[3] = ok ? v[3] + __uint_as_float(eb[1][2]) : v[3];
        pend |= ok ? 0u : (1u << g);
      };
#else
      auto publish4 = [&](int g, const f32x4& v) {
        if (!(SPLIT && ICRL_EARLY_PUBLISH)) return;
        const u64 tg_ = (u64)step << 32;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          GSTORE(mine_e + (size_t)(4 * g + i) * TH4, tg_ | (u64)__float_as_uint(v[i]));
      };
      // ICRL_EARLY_RECV (on since the exchange is XCD-local: 19.6 us per step against 20.8 with the early publish alone; with agent-scope
      // stores across XCDs it was 24.0 against 22.5 — a group that is looked at inside the MFMA stream stalls it when it is late, and
      // the in-order vmcnt makes every look wait for the stores issued since — which is what a launch whose workgroups do NOT share an
      // XCD still pays): the partner's groups fetched the same way — during dW1 tile c the granules of group early_group(c) (which
      // the partner published several tiles ago) are loaded, one tile later they are checked and added to the own (already
      // published) partial; a group that has not arrived is left to the polling pass below (`pend`).
      const u64* const theirs_e = SPLIT ? GP(KARGS()->gx) + ((size_t)((step & 1) * 3 + role) * 2 + (1 - half)) * ((size_t)(KG_ + 5) * TH4) + tid : nullptr;
      u64 eb[4] = {0, 0, 0, 0};
      auto early_issue = [&](int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) eb[i] = __hip_atomic_load(theirs_e + (size_t)(4 * g + i) * TH4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto early_take = [&](int g, f32x4& v) {
        const bool ok = (unsigned)(eb[0] >> 32) == step && (unsigned)(eb[1] >> 32) == step && (unsigned)(eb[2] >> 32) == step && (unsigned)(eb[3] >> 32) == step;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? v[i] + __uint_as_float((unsigned)eb[i]) : v[i];
        pend |= ok ? 0u : (1u << g);
      };
#endif
      // ================= weight gradients: rows 16w.. of dW2 / dW1, columns 16w.. of dWh; K = the 64 rows =================
      {
        f32x4 az[4];   // dz2^T[j = 16w + r][rows 16 js + 4q + e]
        const float* pa = sm + S::DZ2T + (16 * w + r) * ST + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) az[js] = lds128(pa + 16 * js);
        const float* pb = sm + S::H1T + r * ST + 4 * q;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 bh[4];
#pragma unroll
          for (int js = 0; js < 4; ++js) bh[js] = lds128(pb + t * 16 * ST + 16 * js);
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) gW2r[t] = MFMA_F32(az[js][e], bh[js][e], gW2r[t]);
          if (t > 0) publish4(NT1 + t - 1, gW2r[t - 1]);
        }
        float s = 0.f;     // d b2[16w + r] = sum over the rows
#pragma unroll
        for (int js = 0; js < 4; ++js) s += (az[js][0] + az[js][1]) + (az[js][2] + az[js][3]);
        gb2r += quad_rows_sum(s);
      }
      {
        f32x4 ao[4], bh[4];   // dOut^T[o = r][rows], h2^T[j = 16w + r][rows]
        const float* pa = sm + S::DOT + r * ST + 4 * q;
        const float* pb = sm + S::H2T + (16 * w + r) * ST + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) { ao[js] = lds128(pa + 16 * js); bh[js] = lds128(pb + 16 * js); }
        f32x4 acc[4];
#pragma unroll
        for (int js = 0; js < 4; ++js) {
          acc[js] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[js] = MFMA_F32(ao[js][e], bh[js][e], acc[js]);
        }
        publish4(NT1 + 3, gW2r[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) gWhr[i] += (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
        float s = 0.f;      // head bias (wave 0 keeps it): row sums of dOut^T
#pragma unroll
        for (int js = 0; js < 4; ++js) s += (ao[js][0] + ao[js][1]) + (ao[js][2] + ao[js][3]);
        s = quad_rows_sum(s);
        // wave 1 (Gaussian policy): d log_std r = sum of the four per-wave partials
        const float sl = (sm[S::PLS + r] + sm[S::PLS + 16 + r]) + (sm[S::PLS + 32 + r] + sm[S::PLS + 48 + r]);
        gex += w == 0 ? s : ((!DISC && role == 0 && w == 1) ? sl : 0.f);
      }
      if (S::DZ1A) {   // every wave has read h2^T: its storage now takes dz1^T
        lds_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::DZ1T + (16 * t + i) * ST] = dz1c[t][i];
        lds_barrier();
      }
      {
        f32x4 az[4];   // dz1^T[j = 16w + r][rows]
        const float* pa = sm + S::DZ1T + (16 * w + r) * ST + 4 * q;
#pragma unroll
        for (int js = 0; js < 4; ++js) az[js] = lds128(pa + 16 * js);
        const float* pb = sm + xcur + r * ST + 4 * q;     // x^T[k = 16c + r][rows 16 js + 4q + e]
#pragma unroll
        for (int c = 0; c < NT1; ++c) {
          f32x4 bx[4];
#pragma unroll
          for (int js = 0; js < 4; ++js) bx[js] = lds128(pb + c * 16 * ST + 16 * js);
#pragma unroll
          for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) gW1r[c] = MFMA_F32(az[js][e], bx[js][e], gW1r[c]);
          if (c == 0) publish4(NT1 + 4, gWhr); else publish4(c - 1, gW1r[c - 1]);
          if (SPLIT && ICRL_EARLY_PUBLISH && ICRL_EARLY_RECV && !ICRL_ROWS_RAWX) {
            if (c > 0) { const int gp = early_group_of(NT1, c - 1); early_take(gp, gp < NT1 ? gW1r[gp < NT1 ? gp : 0] : (gp < NT1 + 4 ? gW2r[gp - NT1 < 4 && gp >= NT1 ? gp - NT1 : 0] : gWhr)); }
            early_issue(early_group_of(NT1, c));
          }
        }
        if (SPLIT && ICRL_EARLY_PUBLISH && ICRL_EARLY_RECV && !ICRL_ROWS_RAWX) {
          const int gp = early_group_of(NT1, NT1 - 1);
          early_take(gp, gp < NT1 ? gW1r[gp < NT1 ? gp : 0] : (gp < NT1 + 4 ? gW2r[gp - NT1 < 4 && gp >= NT1 ? gp - NT1 : 0] : gWhr));
        }
        float s = 0.f;
#pragma unroll
        for (int js = 0; js < 4; ++js) s += (az[js][0] + az[js][1]) + (az[js][2] + az[js][3]);
        gb1r += quad_rows_sum(s);
        publish4(NT1 - 1, gW1r[NT1 - 1]);
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
    }  // chunks

    if (SPLIT) {
      // ---- partial gradients of this half <-> the other half of the same network.  Slot of element k of thread tid:
      // gx[parity][role][half][k][tid]; KG gradient elements per thread + 5 loss-statistic sums of the book-keeping thread.
      // Gradient registers are published and summed in place, four at a time (no staging copy: the kernel has no registers to
      // spare at AntWall widths), the next group's loads in flight while this one is checked.
      constexpr int KG = 4 * NT1 + 23, NGRP = NT1 + 6;       // groups: W1 tiles, 4 W2 tiles, head, {b1, b2, extra, -}
      const size_t blk = (size_t)(KG + 5) * TH4;
      u64* const mine = GP(KARGS()->gx) + ((size_t)((step & 1) * 3 + role) * 2 + half) * blk + tid;
      const u64* const theirs = GP(KARGS()->gx) + ((size_t)((step & 1) * 3 + role) * 2 + (1 - half)) * blk + tid;
      const u64 tg = (u64)step << 32;
      f32x4 gsc = f32x4{gb1r, gb2r, gex, 0.f};
      auto grp = [&](int g) -> f32x4& { return g < NT1 ? gW1r[g] : (g < NT1 + 4 ? gW2r[g - NT1] : (g == NT1 + 4 ? gWhr : gsc)); };
#if ICRL_ROWS_RAWX
      (void)mine; (void)theirs; (void)tg; (void)KG;
      if (book) gsc[3] = mb_s0;                    // (the book-keeping lane's first loss sum rides in the spare slot of the last group)
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        if (ICRL_EARLY_PUBLISH && g < NT1 + 5) continue;      // (already out, group by group, behind their GEMMs)
        raw_store(xmine + (g * TH4 + tid) * 16, grp(g));
      }
      if (book) raw_store(xmine + NGRP * TH4 * 16, f32x4{mb_s1, mb_s2, mb_s3, mb_s4});
      // every store of this wave has been acknowledged (it is in the L2 the partner reads through, or beyond) -> the wave's flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        if (xcd_local) __builtin_amdgcn_raw_buffer_store_b32(step, grs, xmine + XFLAG + 64 * w, 0, 1);
        else __builtin_amdgcn_raw_buffer_store_b32(step, grs, xmine + XFLAG + 64 * w, 0, 16);
      }
      bool timed_out = false;
      {
        unsigned f = 0;
        int spins = 0;
        while (true) {
          f = __builtin_amdgcn_raw_buffer_load_b32(grs, xtheirs + XFLAG + 64 * w, 0, 16);
          if (f == step) break;
          if (++spins >= (1 << 22)) { timed_out = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      asm volatile("" ::: "memory");
      constexpr int DEP = 7;
      f32x4 ring[DEP];
#pragma unroll
      for (int k = 0; k < DEP; ++k)
        if (k < NGRP) ring[k] = raw_load(xtheirs + (k * TH4 + tid) * 16);
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        f32x4& v = grp(g);
        const f32x4 c = ring[g % DEP];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += c[i];                // own + partner (commutative: both halves agree)
        if (g + DEP < NGRP) ring[g % DEP] = raw_load(xtheirs + ((g + DEP) * TH4 + tid) * 16);
      }
      gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
      if (book) {
        mb_s0 = gsc[3];
        const f32x4 c = raw_load(xtheirs + NGRP * TH4 * 16);
        mb_s1 += c[0]; mb_s2 += c[1]; mb_s3 += c[2]; mb_s4 += c[3];
      }
#elif ICRL_ROWS_REC16
      if (book) gsc[3] = mb_s0;                    // (the book-keeping thread's first loss sum rides in the spare slot of the last group)
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        if (ICRL_EARLY_PUBLISH && g < NT1 + 5) continue;
        rec_pub(g, grp(g));
      }
      if (book) {
        rec_store(rmine + (2 * NGRP) * TH4 * 16, rec_u4{step, __float_as_uint(mb_s1), __float_as_uint(mb_s2), step});
        rec_store(rmine + (2 * NGRP + 1) * TH4 * 16, rec_u4{step, __float_as_uint(mb_s3), __float_as_uint(mb_s4), step});
      }
      bool timed_out = false;
      constexpr int DEP = 4;
      rec_u4 ring[DEP][2];
      constexpr RemGroups<NT1> REM = RemGroups<NT1>::make(SPLIT && ICRL_EARLY_PUBLISH && ICRL_EARLY_RECV);
      auto issue = [&](rec_u4 (&b)[2], int g) { b[0] = rec_load(rtheirs + (2 * g) * TH4 * 16); b[1] = rec_load(rtheirs + (2 * g + 1) * TH4 * 16); };
      auto add_to = [&](f32x4& v, const rec_u4 (&c)[2], bool ok) {
        v[0] = ok ? v[0] + __uint_as_float(c[0][1]) : v[0]; v[1] = ok ? v[1] + __uint_as_float(c[0][2]) : v[1];
        v[2] = ok ? v[2] + __uint_as_float(c[1][1]) : v[2]; v[3] = ok ? v[3] + __uint_as_float(c[1][2]) : v[3];
      };
#pragma unroll
      for (int k = 0; k < DEP; ++k)
        if (k < REM.n) issue(ring[k], REM.v[k]);
#pragma unroll
      for (int k = 0; k < REM.n; ++k) {
        const int g = REM.v[k];
        rec_u4 (&c)[2] = ring[k % DEP];
        const bool ok = rec_ok(c[0], c[1]);
        add_to(grp(g), c, ok);
        pend |= ok ? 0u : (1u << g);
        if (k + DEP < REM.n) issue(c, REM.v[k + DEP]);
      }
      if (__any(pend != 0u)) {
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          if (!__any((pend >> g) & 1u)) continue;
          const bool mine_pending = (pend >> g) & 1u;
          rec_u4 c[2] = {rec_u4{0, 0, 0, 0}, rec_u4{0, 0, 0, 0}};
          bool ok = !mine_pending;
          for (int spins = 0; spins < (1 << 22) && !timed_out; ++spins) {
            if (!ok) { issue(c, g); ok = rec_ok(c[0], c[1]); }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins + 1 == (1 << 22)) timed_out = true;
          }
          if (mine_pending && ok) add_to(grp(g), c, true);
        }
      }
      gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
      if (book) {
        mb_s0 = gsc[3];
        rec_u4 c[2] = {rec_u4{0, 0, 0, 0}, rec_u4{0, 0, 0, 0}};
        for (int spins = 0; spins < (1 << 22) && !timed_out; ++spins) {
          issue(c, NGRP);
          if (rec_ok(c[0], c[1])) break;
          if (spins + 1 == (1 << 22)) timed_out = true;
        }
        mb_s1 += __uint_as_float(c[0][1]); mb_s2 += __uint_as_float(c[0][2]); mb_s3 += __uint_as_float(c[1][1]); mb_s4 += __uint_as_float(c[1][2]);
      }
#else
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        if (ICRL_EARLY_PUBLISH && g < NT1 + 5) continue;      // (already out, group by group, behind their GEMMs)
        const f32x4 v = grp(g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * g + i < KG)
            GSTORE(mine + (size_t)(4 * g + i) * TH4, tg | (u64)__float_as_uint(v[i]));
      }
      if (book) {
        const float ms[5] = {mb_s0, mb_s1, mb_s2, mb_s3, mb_s4};
#pragma unroll
        for (int k = 0; k < 5; ++k)
          GSTORE(mine + (size_t)(KG + k) * TH4, tg | (u64)__float_as_uint(ms[k]));
      }
      // Fast pass, straight-line on purpose (any loop or branch between a load and its use makes the compiler wait for ALL
      // outstanding loads): DEP groups in flight, a group is added where all four of its granules carry this step's tag; the
      // rest is remembered per lane and fetched by the polling loop below.  (Measured: the exchange is throughput-bound, not
      // latency-bound — ~14 k granules stored and ~14 k loaded per workgroup and step at about one 8-byte granule per cycle and
      // CU; waiting before the first pass changes nothing.)
      bool timed_out = false;
      constexpr int DEP = 4;
      u64 ring[DEP][4];
      constexpr RemGroups<NT1> REM = RemGroups<NT1>::make(SPLIT && ICRL_EARLY_PUBLISH && ICRL_EARLY_RECV);
      auto issue = [&](u64 (&b)[4], int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          b[i] = (4 * g + i < KG) ? __hip_atomic_load(theirs + (size_t)(4 * g + i) * TH4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tg;
      };
      auto arrived = [&](const u64 (&b)[4]) -> bool {
        return (unsigned)(b[0] >> 32) == step && (unsigned)(b[1] >> 32) == step && (unsigned)(b[2] >> 32) == step && (unsigned)(b[3] >> 32) == step;
      };
#pragma unroll
      for (int k = 0; k < DEP; ++k)
        if (k < REM.n) issue(ring[k], REM.v[k]);
#pragma unroll
      for (int k = 0; k < REM.n; ++k) {
        const int g = REM.v[k];
        u64 (&c)[4] = ring[k % DEP];
        const bool ok = arrived(c);
        f32x4& v = grp(g);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? v[i] + __uint_as_float((unsigned)c[i]) : v[i];   // own + partner (commutative: both halves agree)
        pend |= ok ? 0u : (1u << g);
        if (k + DEP < REM.n) issue(c, REM.v[k + DEP]);
      }
      if (__any(pend != 0u)) {
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          if (!__any((pend >> g) & 1u)) continue;
          const bool mine_pending = (pend >> g) & 1u;
          u64 c[4] = {0, 0, 0, 0};
          bool ok = !mine_pending;
          for (int spins = 0; spins < (1 << 22) && !timed_out; ++spins) {
            if (!ok) { issue(c, g); ok = arrived(c); }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins + 1 == (1 << 22)) timed_out = true;
          }
          if (mine_pending && ok) {
            f32x4& v = grp(g);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += __uint_as_float((unsigned)c[i]);
          }
        }
      }
      gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
      if (book) {
        float ms[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          u64 v = 0;
          for (int spins = 0; spins < (1 << 22) && !timed_out; ++spins) {
            v = __hip_atomic_load(theirs + (size_t)(KG + k) * TH4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == step) break;
            if (spins + 1 == (1 << 22)) timed_out = true;
          }
          ms[k] = __uint_as_float((unsigned)v);
        }
        mb_s0 += ms[0]; mb_s1 += ms[1]; mb_s2 += ms[2]; mb_s3 += ms[3]; mb_s4 += ms[4];
      }
#endif
      if (timed_out) sm[S::MISC + 13] = 1.f;       // reported through the status word like a timed-out norm exchange
    }

    // entropy term of the Gaussian policy loss: d(ent_coef * -mean(H)) / d log_std = -ent_coef
    if (!DISC && role == 0 && w == 1 && r < A) gex += -ent_coef;

    // ================= global gradient norm: this wave's partial sum of squares -> its own 8-byte granule =================
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NT1; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) ss = fmaf(gW1r[c][i], gW1r[c][i], ss);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) ss = fmaf(gW2r[c][i], gW2r[c][i], ss);
#pragma unroll
    for (int i = 0; i < 4; ++i) ss = fmaf(gWhr[i], gWhr[i], ss);
    {
      const float sb = fmaf(gb1r, gb1r, gb2r * gb2r) + (ex_g >= 0 ? gex * gex : 0.f);
      ss += q == 0 ? sb : 0.f;
    }
    ss = wave_sum_fast(ss);
    if (lane == 0) {
      bool want_stop = false;
      float mean_kl = 0.f;
      const bool last_mb = (ps.nb_flags >> NB_LAST) & 1;
      const int epoch = ps.nb_flags >> NB_EPOCH;
      if (book && role == 0) {   // the early-stop decision rides on the granule of the policy workgroup's wave 3
        if ((ps.nb_flags >> NB_FIRST) & 1) kl_sum = 0.f;
        kl_sum += mb_s3 * inv_nb;
        if (last_mb) {
          mean_kl = kl_sum * inv_n_mb;
          const TrainArgs* k_ = KARGS();
          if (k_->hp.use_target_kl && mean_kl > 1.5f * k_->hp.target_kl) { want_stop = true; early_stop_epoch = epoch; }
        }
      }
      const unsigned tag = step | (want_stop ? 0x80000000u : 0u);
      GSTORE(xch + half * 32 + (step & 1) * 16 + role * 4 + w, ((u64)tag << 32) | (u64)__float_as_uint(ss));
      if (book) {
        ++steps_done;
        if (role == 0) {
          float ent = 0.f;
          if (DISC) ent = mb_s4 * inv_nb;
          else ent = sm[S::MISC + 22];
          const float entropy_loss = -ent;
          const float pl = (-(mb_s0 * inv_nb) + nu * (mb_s1 * inv_nb)) * __builtin_amdgcn_rcpf(1.f + nu);
          st_ent += entropy_loss; st_pg += pl; st_cf += mb_s2 * inv_nb;
          last_loss = pl + ent_coef * entropy_loss;
          if (last_mb) { float* stats = KARGS()->stats; stats[32 + epoch] = mean_kl; stats[7] = mean_kl; }
        } else {
          const float vl = mb_s0 * inv_nb;
          st_vl += vl;
          last_loss = vl;
        }
      }
    }
    STAMP(4)   // gradient norm + publish
    // ---- while the granules travel: stage the next minibatch (rows -> the other X^T buffer, advantage statistics)
    const int xnext = S::XDB ? (xcur == S::XT0 ? S::XT1 : S::XT0) : xcur;
#if ICRL_ROWS_EARLY_POLL
    u64 v_first = 0;
    if (tid < 12) v_first = __hip_atomic_load(xch + half * 32 + (step & 1) * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    commit_rows(xnext);
    const int nb_next = __builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK;
    stats_partials(nb_next);
    xcur = xnext;
    if (tid < 12) {
      u64 v = 0;
      int spins = 0;
      bool ok = false;
      const u64* const slot = xch + half * 32 + (step & 1) * 16 + tid;     // (the three networks of the same half)
#if ICRL_ROWS_EARLY_POLL
      v = v_first;
      if ((unsigned)((v >> 32) & 0x7fffffffu) == step) { ok = true; spins = 1 << 24; }
#endif
      while (spins < (1 << 24)) {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)((v >> 32) & 0x7fffffffu) == step) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
        ++spins;
      }
      sm[S::MISC + tid] = __uint_as_float((unsigned)(v & 0xffffffffu));
      if (tid == 3) sm[S::MISC + 12] = (v >> 63) ? 1.f : 0.f;     // granule 3 = (policy, wave 3) carries the stop flag
      if (!ok) sm[S::MISC + 13] = 1.f;
    }
    lds_barrier();   // (B3) norm partials, next minibatch and its statistics visible
    STAMP(5)   // staging + granule wait
    float total;
    {
      const f32x4 n0 = lds128(sm + S::MISC), n1 = lds128(sm + S::MISC + 4), n2 = lds128(sm + S::MISC + 8), fl = lds128(sm + S::MISC + 12);
      total = ((n0[0] + n0[1]) + (n0[2] + n0[3])) + (((n1[0] + n1[1]) + (n1[2] + n1[3])) + ((n2[0] + n2[1]) + (n2[2] + n2[3])));
      stop = fl[0] != 0.f;
      if (fl[1] != 0.f) { status = 1; stop = true; }
    }
    total = __builtin_amdgcn_sqrtf(total);
    float coef = max_grad_norm * __builtin_amdgcn_rcpf(total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    read_stats(nb_next > 0 ? nb_next : 2);

    // ================= Adam (torch.optim.Adam, single-tensor form) on register-resident weights and moments =================
    {   // unconditional: after a timed-out exchange (status != 0) the launch ends with the status word set and the host raises
      const float step_size = ps.step_size, inv_bc2_sqrt = ps.inv_bc2_sqrt;
      const float epsf = adam_epsf;
      const float omw1 = 1.f - w1, b2f_ = adam_b2f;
      const float cw1 = coef * w1, c2w2 = (coef * coef) * w2;
      // four elements at a time, stage by stage (independent chains keep the transcendental unit and the FMA pipe busy; one
      // wave per SIMD has nobody else to cover a dependent v_sqrt -> v_rcp chain)
      auto adam4 = [&](const f32x4& g, float* mA4, float* vA4, f32x4& p) {
        f32x4 m, v, d;
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = acc_get(mA4[i]); v[i] = acc_get(vA4[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = fmaf(cw1, g[i], omw1 * m[i]); v[i] = fmaf(c2w2, g[i] * g[i], b2f_ * v[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = fmaf(__builtin_amdgcn_sqrtf(v[i]), inv_bc2_sqrt, epsf);   // v_sqrt_f32 / v_rcp_f32: 1 ulp each
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = fmaf(-step_size, m[i] * __builtin_amdgcn_rcpf(d[i]), p[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc_set(mA4[i], m[i]); acc_set(vA4[i], v[i]); }
      };
      // pad elements (k >= obs, o >= n_out) have g = m = v = p = 0 and stay 0: no masks needed
#pragma unroll
      for (int c = 0; c < NT1; ++c) { f32x4 p_ = load_own_w1(c); adam4(gW1r[c], mA + 4 * c, vA + 4 * c, p_); store_w1(c, p_); }
#pragma unroll
      for (int c = 0; c < 4; ++c) { f32x4 p_ = load_own_w2(c); adam4(gW2r[c], mA + E_W2 + 4 * c, vA + E_W2 + 4 * c, p_); store_w2(c, p_); }
      { f32x4 p_ = load_own_wh(); adam4(gWhr, mA + E_WH, vA + E_WH, p_); store_wh(p_); }
      {   // b1 / b2 / extra entry: identical arithmetic in the four q lanes, lane q == 0 stores (elements E_B1, E_B2, E_EX are adjacent)
        f32x4 g_ = f32x4{gb1r, gb2r, ex_g >= 0 ? gex : 0.f, 0.f}, p_ = f32x4{sm[S::B1 + jb], sm[S::B2 + jb], sm[ex_s], 0.f};
        float m3[4], v3[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) { m3[i] = mA[E_B1 + i]; v3[i] = vA[E_B1 + i]; }
        acc_put(m3[3], 0.f); acc_put(v3[3], 0.f);
        adam4(g_, m3, v3, p_);
#pragma unroll
        for (int i = 0; i < 3; ++i) { mA[E_B1 + i] = m3[i]; vA[E_B1 + i] = v3[i]; }
        if (q == 0) { sm[S::B1 + jb] = p_[0]; sm[S::B2 + jb] = p_[1]; sm[ex_s] = p_[2]; }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this lane's log_std store has landed before refresh_gauss re-reads it
      refresh_gauss();
    }
    lds_barrier();   // (B4) updated weights visible
    STAMP(6)   // Adam
  }  // optimiser steps

  __syncthreads();
  // ---- write back weights, moments, statistics.  The pointers / offsets are re-read from the kernel-argument segment here
  // instead of being kept in scalar registers across the whole optimisation loop.  SPLIT: the two halves are replicas; half 0 writes.
  if (!SPLIT || half == 0) {
  const TrainArgs* kw = ka;
  asm volatile("" : "+s"(kw));
  const TrainArgs& a = *kw;
  const PolLayout& L = a.L;
  const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
  const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
  const int gbh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);
  (void)gbh;
#pragma unroll
  for (int c = 0; c < NT1; ++c) {
    const f32x4 pv = load_own_w1(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * w + 4 * q + i, k = 16 * c + r;
      if (k < O) { a.params[gW1 + j * O + k] = pv[i]; a.exp_avg[gW1 + j * O + k] = acc_get(mA[4 * c + i]); a.exp_avg_sq[gW1 + j * O + k] = acc_get(vA[4 * c + i]); }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 pv = load_own_w2(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * w + 4 * q + i, k = 16 * c + r;
      a.params[gW2 + j * HD + k] = pv[i];
      a.exp_avg[gW2 + j * HD + k] = acc_get(mA[E_W2 + 4 * c + i]);
      a.exp_avg_sq[gW2 + j * HD + k] = acc_get(vA[E_W2 + 4 * c + i]);
    }
  }
  {
    const f32x4 pv = load_own_wh();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = 4 * q + i, j = 16 * w + r;
      if (o < n_out) { a.params[gWh + o * HD + j] = pv[i]; a.exp_avg[gWh + o * HD + j] = acc_get(mA[E_WH + i]); a.exp_avg_sq[gWh + o * HD + j] = acc_get(vA[E_WH + i]); }
    }
  }
  if (q == 0) {
    a.params[gb1 + jb] = sm[S::B1 + jb]; a.exp_avg[gb1 + jb] = acc_get(mA[E_B1]); a.exp_avg_sq[gb1 + jb] = acc_get(vA[E_B1]);
    a.params[gb2 + jb] = sm[S::B2 + jb]; a.exp_avg[gb2 + jb] = acc_get(mA[E_B2]); a.exp_avg_sq[gb2 + jb] = acc_get(vA[E_B2]);
    if (ex_g >= 0) { a.params[ex_g] = sm[ex_s]; a.exp_avg[ex_g] = acc_get(mA[E_EX]); a.exp_avg_sq[ex_g] = acc_get(vA[E_EX]); }
  }
  if (tid == 0 && prof) {
    for (int k = 0; k < 7; ++k) {
      const int slot = 12 + 7 * role + k;
      if (slot < 32) a.stats[slot] = (float)((double)ph[k] / (double)(a.n_steps > 0 ? a.n_steps : 1));   // (full runs only)
    }
  }
  if (book) {
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
}

template <int NT1, bool DISC, bool SPLIT>
__global__ void __launch_bounds__(TH4) ppo_train_rows_kernel(TrainArgs a, int packed) {
  int run = 0, j = (int)blockIdx.x;
  if (packed && !packed_slot(SPLIT ? 6 : 3, 1, run, j)) return;
  ppo_train_rows_body<NT1, DISC, SPLIT, false>(a, (const TrainArgs*)__builtin_amdgcn_kernarg_segment_ptr(), j);
}

// several independent runs in ONE launch: the packed 1-D grid of ppo_common.h (a run's workgroups on one XCD), or grid (3 or 6, n_runs)
// with run = blockIdx.y when that many workgroups are not resident at once
template <int NT1, bool DISC, bool SPLIT>
__global__ void __launch_bounds__(TH4) ppo_train_rows_batch_kernel(const TrainArgs* __restrict__ runs, int n_runs, int packed) {
  int run = (int)blockIdx.y, j = (int)blockIdx.x;
  if (packed && !packed_slot(SPLIT ? 6 : 3, n_runs, run, j)) return;
  const TrainArgs* const ka = as_global(runs + run);
  ppo_train_rows_body<NT1, DISC, SPLIT, true>(*ka, ka, j);
}

template <int NT1, bool DISC, bool SPLIT>
static int launch_rows_batch(const TrainArgs* d_args, int n_runs, hipStream_t s) {
  const size_t bytes = ICRL_ROWS_STATIC_LDS ? 0 : (size_t)SmemR<NT1>::TOTAL * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)ppo_train_rows_batch_kernel<NT1, DISC, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  const int pg = packed_grid(SPLIT ? 6 : 3, n_runs);
  hipLaunchKernelGGL((ppo_train_rows_batch_kernel<NT1, DISC, SPLIT>), pg ? dim3(pg) : dim3(SPLIT ? 6 : 3, n_runs), dim3(TH4), bytes, s, d_args, n_runs, pg ? 1 : 0);
  return (int)hipGetLastError();
}

int launch_train_rows_batch(const TrainArgs* d_args, int n_runs, int nt1, bool discrete, bool split, hipStream_t s) {
  if (split) {
    if (nt1 <= 4) return discrete ? launch_rows_batch<4, true, true>(d_args, n_runs, s) : launch_rows_batch<4, false, true>(d_args, n_runs, s);
    if (nt1 <= 8) return discrete ? launch_rows_batch<8, true, true>(d_args, n_runs, s) : launch_rows_batch<8, false, true>(d_args, n_runs, s);
    return fail("update (row-owning waves): obs_dim tiles %d > 8", nt1);
  }
  if (nt1 <= 2) return discrete ? launch_rows_batch<2, true, false>(d_args, n_runs, s) : launch_rows_batch<2, false, false>(d_args, n_runs, s);
  if (nt1 <= 4) return discrete ? launch_rows_batch<4, true, false>(d_args, n_runs, s) : launch_rows_batch<4, false, false>(d_args, n_runs, s);
  if (nt1 <= 8) return discrete ? launch_rows_batch<8, true, false>(d_args, n_runs, s) : launch_rows_batch<8, false, false>(d_args, n_runs, s);
  return fail("update (row-owning waves): obs_dim tiles %d > 8", nt1);
}

template <int NT1, bool DISC, bool SPLIT>
static int launch_rows(const TrainArgs& a, hipStream_t s) {
  static_assert(SmemR<NT1>::TOTAL * sizeof(float) <= 160 * 1024, "LDS budget");
  const size_t bytes = ICRL_ROWS_STATIC_LDS ? 0 : (size_t)SmemR<NT1>::TOTAL * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)ppo_train_rows_kernel<NT1, DISC, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  TrainArgs arg = a;
  return launch_update_single(ppo_train_rows_kernel<NT1, DISC, SPLIT>, SPLIT ? 6 : 3, dim3(TH4), bytes, s, arg);
}

int launch_train_rows(const TrainArgs& a, int nt1, bool discrete, bool split, hipStream_t s) {
  if (split) {       // two workgroups per network (a.gx set, the chunk plan holds two entries per step)
    if (nt1 <= 4) return discrete ? launch_rows<4, true, true>(a, s) : launch_rows<4, false, true>(a, s);
    if (nt1 <= 8) return discrete ? launch_rows<8, true, true>(a, s) : launch_rows<8, false, true>(a, s);
    return fail("update (row-owning waves): obs_dim tiles %d > 8", nt1);
  }
  if (nt1 <= 2) return discrete ? launch_rows<2, true, false>(a, s) : launch_rows<2, false, false>(a, s);
  if (nt1 <= 4) return discrete ? launch_rows<4, true, false>(a, s) : launch_rows<4, false, false>(a, s);
  if (nt1 <= 8) return discrete ? launch_rows<8, true, false>(a, s) : launch_rows<8, false, false>(a, s);
  return fail("update (row-owning waves): obs_dim tiles %d > 8", nt1);
}

}  // namespace icrl
