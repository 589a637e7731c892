// PPO-Lagrangian update, FOUR workgroups per network at WIDE observations, TWO row tiles per wave in ONE pass (obs_dim 65..128, minibatches of
// 65..128 rows: AntWall / AntWallBroken with the reference's batch size 128 — BASELINE configs[2], [4]) — gfx950.
//
// ref: stable_baselines3/ppo_lag/ppo_lag.py:196-299, common/buffers.py:594-627, common/policies.py:752-767,
//      common/distributions.py:143-171,274-288, torch.optim.Adam, clip_grad_norm_  (same contract as ppo_train_rows.hip).
//
// ppo_train_quarters.hip walks a 128-row minibatch as two 64-row chunks, 16 rows of each per workgroup, one after the other: every LDS hand-off
// of the wave quad, every barrier and every operand fetch of the weights is paid twice per optimiser step, and each pass is one dependent MFMA
// chain per wave.  Here part p takes rows 16 p .. 16 p + 15 of BOTH chunks at once (the chunk plan with two entries per step that
// ppo_train_rows.hip's two-workgroup form uses): a wave carries two independent row tiles through its feature tile of every layer — the weight
// operands are fetched once for both, the two MFMA chains interleave, and there are three hand-offs and three barriers per step instead of six and
// five; the weight-gradient GEMMs have K = 32.  Everything else (parameter ownership, exchange of the four partial gradients as raw 16-byte stores
// + one flag per wave, (own + partner) + (the other pair), replicated norm / Adam, part 0 writes back) is ppo_train_quarters.hip's.
// LDS: the images of 32 rows leave no room for a second X^T buffer — the next minibatch is staged between the norm barrier and the barrier behind
// Adam, when no wave reads X^T any more — and the row-major image of dz2 lives in h1's (dead by then).
//
// Built with -ffp-contract=off; FMA is used only where written (fmaf / MFMA).
#include "ppo_common.h"

#ifndef ICRL_QW2_EARLY_PUBLISH
#define ICRL_QW2_EARLY_PUBLISH 1
#endif
#ifndef ICRL_QW2_DEP
#define ICRL_QW2_DEP 4
#endif

// the Adam moments as plain values (the register allocator places them) instead of pinned to accumulation registers with explicit moves
// (ppo_train_rows.hip's scheme): the step loop loses ~150 instructions net — 12.30 -> 12.07 us per step (two alternating rounds), no scratch traffic in the loop
#ifndef ICRL_QW2_MOMENTS_PLAIN
#define ICRL_QW2_MOMENTS_PLAIN 1
#endif
// the loss tail's per-row operands and dH2's A operand fetched from LDS before the head hand-off (P3) instead of behind it (ppo_train_halves.hip:
// 6.00 -> 5.98 us there); here 12.30 vs 12.31 — nothing; off
#ifndef ICRL_QW2_LOSS_PRELOAD
#define ICRL_QW2_LOSS_PRELOAD 0
#endif
// the three hand-offs inside the wave quad as WORKGROUP BARRIERS: here the quad is the whole workgroup (ppo_train_halves.hip needs LDS flags — its waves 4..7
// are parked at another barrier meanwhile), so `s_waitcnt lgkmcnt(0); s_barrier` replaces a flag store plus a polling loop of three LDS loads per look:
// 11.68 -> 11.53 us per step (three alternating rounds)
#ifndef ICRL_QW2_QUAD_BARRIER
#define ICRL_QW2_QUAD_BARRIER 1
#endif
#ifndef ICRL_QW_STATIC_LDS
#define ICRL_QW_STATIC_LDS 0
#endif

namespace icrl {

#if ICRL_QW2_MOMENTS_PLAIN
#define acc_put(slot, v) ((slot) = (v))
#define acc_set(slot, v) ((slot) = (v))
#define acc_get(slot) (slot)
#endif

constexpr int THQ2 = 256;  // 4 waves, one per SIMD
constexpr int STX = 36;    // row stride of the [feature][row] images (32 rows + 4: conflict-free ds_read_b128 and column stores)
constexpr int SRX = 72;    // row stride of the [row][feature] images
constexpr int SAX = 24;    // row stride of the per-row action block and of the transposed head weights

template <int NT1>
struct SmemQ2 {  // offsets in floats (multiples of 4)
  static constexpr int O16 = 16 * NT1, SX = O16 + 8;
  // The [feature][row] / [row][feature] images come FIRST: a ds instruction reaches 64 KB beyond its address register, and the compiler keeps one
  // register per distinct address whose constant part does not fit (it held ~60 of them in accumulation registers and scratch, reloaded one by one
  // in front of the stores of every phase).  What lies beyond 64 KB is addressed through a few opaque per-pattern bases (`opq` below).
  static constexpr int XT = 0;                 // [16 NT1][STX] x^T of this part's rows: XT[k][16 tile + row]
  static constexpr int H1T = XT + O16 * STX;   // [64][STX] h1^T
  static constexpr int H2T = H1T + HD * STX;
  static constexpr int DZ1T = H2T + HD * STX;
  static constexpr int DZ2T = DZ1T + HD * STX;
  static constexpr int DOT = DZ2T + HD * STX;  // [16][STX] d loss / d head output, transposed
  static constexpr int H1R = DOT + 16 * STX;   // [32][SRX] h1, row-major (B operand of layer 2 for the other three waves)
  static constexpr int DZ2R = H1R;             // [32][SRX] dz2, row-major (B operand of dH1) — in h1's storage: a wave writes dz2 behind the head hand-off (P3),
                                               // which every wave signals after its layer-2 reads of h1 have returned
  static constexpr int HPX = H1R + 32 * SRX;   // [2][4][64][4] head partial tiles of the four waves, per row tile
  static constexpr int ACT = HPX + 2048;       // [32][SAX] actions of this part's rows
  static constexpr int OLP = ACT + 32 * SAX;   // [32] old log-prob | old value
  static constexpr int ADR = OLP + 32;         // [32] raw reward advantage | return
  static constexpr int ADC = ADR + 32;         // [32] raw cost advantage
  static constexpr int PST = ADC + 32;         // [2][8] loss statistics of the row tiles
  static constexpr int PLS = PST + 16;         // [2][16] d log_std partial sums of the row tiles
  static constexpr int MISC = PLS + 32;        // [64]: 0..8 advantage-statistics partials, 12 stop, 13 timed out, 14 one XCD, 22 entropy,
                                               //       24..47 norm partials [role][8], 48..51 quad flags, 62 / 63 scratch words
  static constexpr int GAU = MISC + 64;        // [3][16] per-action 1/var, 0.5/var, log(sd) + log(sqrt(2 pi))
  static constexpr int B1 = GAU + 48;
  static constexpr int B2 = B1 + HD;
  static constexpr int BH = B2 + HD;
  static constexpr int LS = BH + 16;
  static constexpr int WH = LS + 16;           // [16][SH]
  static constexpr int WHT = WH + 16 * SH;     // [64][SAX]  WHT[j][o] = WH[o][j]
  static constexpr int W2 = WHT + HD * SAX;    // [64][SH]
  static constexpr int W2T = W2 + HD * SH;     // [64][SH]  W2T[k][j] = W2[j][k]
  static constexpr int W1 = W2T + HD * SH;     // [64][SX]
  static constexpr int TOTAL = W1 + HD * SX;
};

#define KARGS() ([&]() { const TrainArgs* k_ = ka; asm volatile("" : "+s"(k_)); return k_; }())

// PROF: the diagnostic phase timers (hp._pad & 1) as a compile-time variant (ppo_train_halves.hip: as a run-time flag they cost every launch ~2.5 %)
template <int NT1, bool DISC, int OBS, bool PROF>
__device__ __forceinline__ void ppo_train_quarters2_body(const TrainArgs& a, const TrainArgs* const ka, const int slot_j) {
  using S = SmemQ2<NT1>;
  constexpr int SX = S::SX;
  static_assert(OBS == 0 || (OBS > 16 * (NT1 - 1) && OBS <= 16 * NT1), "OBS names the observation width of an NT1-tile instantiation");
#if ICRL_QW_STATIC_LDS
  __shared__ __attribute__((aligned(16))) float sm[S::TOTAL];
#else
  extern __shared__ __attribute__((aligned(16))) float sm[];      // dynamic: with a static array the folded offsets let the optimiser hoist more addresses than the register file holds (ppo_train_rows.hip)
#endif
  // fault injection for the tests (hp._pad & 64): the last workgroup of the run leaves at once — every wait of the others is bounded, the launch ENDS
  // with the status word set and the host raises
  if ((a.hp._pad & 64) && slot_j == 11) return;
  const int role = slot_j % 3;   // 0 policy, 1 reward critic, 2 cost critic
  const int part = slot_j / 3;   // rows 16 part .. 16 part + 15 of both 64-row chunks of a minibatch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // forward / activation backward: feature tile; weight gradients / Adam: parameter row block
  const int qp0 = w ^ 1, qp1 = w ^ 2, qp2 = w ^ 3;          // the other three waves of the quad
  const int r = lane & 15, q = lane >> 4;
  const int O = OBS > 0 ? OBS : a.L.O, A = a.L.A;
  const int n_out = role == 0 ? A : 1;
  const int T = a.buf.T, N = a.buf.N;
  const float nu = a.nu[0];
  const int n_steps = a.n_steps;
  const PlanStep* __restrict__ const plan_steps = a.plan_steps;
  const PlanChunk* __restrict__ const plan_chunks = a.plan_chunks;
  const int* __restrict__ const perms = a.perms;
  const float* const p_s0 = role == 0 ? a.buf.log_probs : (role == 1 ? a.buf.reward_values : a.buf.cost_values);
  const float* const p_s1 = role == 0 ? a.buf.reward_advantages : (role == 1 ? a.buf.reward_returns : a.buf.cost_returns);
  const float* const p_s2 = a.buf.cost_advantages;
  const float* const p_obs = a.buf.observations;
  const float* const p_act = a.buf.actions;
  const int AS = a.buf.act_store;

  // ---- Adam ownership (ppo_train_rows.hip): wave w owns W1 / W2 rows 16 w .. (element j = 16 w + 4 q + i, k = 16 c + r), head-weight columns
  // 16 w .. (element o = 4 q + i, j = 16 w + r), b1 / b2 entries 16 w + r (replicated over q, lane q == 0 stores); wave 0: head bias r,
  // wave 1: log_std r.  Master weights in LDS (the operand copies ARE the master values), moments in accumulation registers.
  constexpr int E_W2 = 4 * NT1, E_WH = E_W2 + 16, E_B1 = E_WH + 4, E_B2 = E_B1 + 1, E_EX = E_B2 + 1, NEL = E_EX + 1;
  float mA[NEL], vA[NEL];                      // AGPR-resident (acc_put / acc_get)
  f32x4 gW1r[NT1], gW2r[4], gWhr;
  const int jb = 16 * w + r;
  // per-pattern bases of what lies beyond the first 64 KB of the LDS: opaque to the optimiser, so that every access is base + immediate
  auto opq = [](int v) { asm volatile("" : "+v"(v)); return v; };
  const int o_w1a = opq(S::W1 + (16 * w + r) * SX + 4 * q);        // layer 1: A operand rows of W1
  const int o_w1o = opq(S::W1 + (16 * w + 4 * q) * SX + r);        // Adam: own elements of W1
  const int o_w2a = opq(S::W2 + (16 * w + r) * SH + 4 * q);        // layer 2 / dH1: A operand rows of W2 (W2T: + S::W2T - S::W2)
  const int o_w2o = opq(S::W2 + (16 * w + 4 * q) * SH + r);        // Adam: own elements of W2
  const int o_w2t = opq(S::W2T + r * SH + 16 * w + 4 * q);         // Adam: own elements of W2T
  const int o_wha = opq(S::WH + r * SH + 16 * w + 4 * q);          // head: A operand
  const int o_who = opq(S::WH + (4 * q) * SH + 16 * w + r);        // Adam: own elements of WH
  const int o_wht = opq(S::WHT + (16 * w + r) * SAX + 4 * q);      // dH2: A operand; Adam: own elements of WHT
  const int o_bia = opq(S::B1 + 16 * w + 4 * q);                   // b1 (b2: + HD) of the own feature tile
  const int o_bo = opq(S::B1 + jb);                                // Adam: own entry of b1 (b2: + HD)
  const int o_gau = opq(S::GAU + 4 * q);
  const int o_bh = opq(S::BH + 4 * q);
  const int o_hpx = opq(S::HPX + lane * 4);
  const int o_act = opq(S::ACT + r * SAX + 4 * q);                 // loss: actions of row r of a tile
  const int o_sid = opq(S::OLP + r);                               // loss: per-row side data (ADR: + 32, ADC: + 64)
  const int o_msc = opq(S::MISC);
  float gb1r = 0.f, gb2r = 0.f, gex = 0.f;
  int ex_g = -1, ex_s = S::MISC + 63;
  {
    const PolLayout& L = a.L;
    const int gbh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);
    if (w == 0 && r < n_out) { ex_g = gbh + r; ex_s = S::BH + r; }
    if (w == 1 && !DISC && role == 0 && r < A) { ex_g = L.log_std + r; ex_s = S::LS + r; }
  }
  auto store_w1 = [&](int c, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[o_w1o + i * SX + 16 * c] = v[i];
  };
  auto load_own_w1 = [&](int c) -> f32x4 {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[o_w1o + i * SX + 16 * c];
    return v;
  };
  auto store_w2 = [&](int c, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[o_w2o + i * SH + 16 * c] = v[i];
    *reinterpret_cast<f32x4*>(sm + o_w2t + 16 * c * SH) = v;
  };
  auto load_own_w2 = [&](int c) -> f32x4 { return lds128(sm + o_w2t + 16 * c * SH); };
  auto store_wh = [&](const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[o_who + i * SH] = v[i];
    *reinterpret_cast<f32x4*>(sm + o_wht) = v;
  };
  auto load_own_wh = [&]() -> f32x4 { return lds128(sm + o_wht); };
  for (int i = tid; i < S::TOTAL; i += THQ2) sm[i] = 0.f;
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
      sm[o_bo] = a.params[gb1 + jb]; sm[o_bo + HD] = a.params[gb2 + jb];
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
  u64* const xch0 = a.xch;                                 // XCD words of all twelve workgroups, part 0's norm granules
  u64* const gxp = a.gx;
  // this part's norm granules: [step parity][role][8] (waves 0..3 used), 512 B per part
  u64* const nx = part == 0 ? xch0 : reinterpret_cast<u64*>(reinterpret_cast<char*>(gxp) + ICRL_PPO_SPLIT_BYTES - 512 * part);

  // ---- row stream: the 2 x 16 rows of this part are staged by the 256 threads, 16 per row and two rows each — row gb_row of chunk 0 and of
  // chunk 1 (ppo_train_pairs.hip for the rules the index / row loads follow: unconditional, clamped, untouched until consumed)
  const int gb_row = tid >> 4, gpart = tid & 15;
  const int gpos = 16 * part + gb_row;           // position of those rows in their 64-row chunks
  const int stid = tid - 64;                     // advantage statistics: row stid of the minibatch on waves 1, 2
  auto ld_step = [&](int i) -> int4 {
    asm volatile("" : "+v"(i));
    return *reinterpret_cast<const int4*>(plan_steps + i);
  };
  auto ld_chunk = [&](int g) -> int2 {           // the plan holds two entries per step (an absent second chunk as 0 rows)
    asm volatile("" : "+v"(g));
    return *reinterpret_cast<const int2*>(plan_chunks + g);
  };
  auto chunk_idx = [&](const int2& c) -> int { return perms[c.x + (gpos < c.y ? gpos : 0)]; };      // {perm_base, rows}
  auto stat_idx = [&](const int4& p) -> int {       // row stid of that step's minibatch (policy role); p.z = nb_flags, p.w = perm_base
    return (role == 0 && stid >= 0 && stid < (p.z & NB_MASK)) ? perms[p.w + stid] : -1;
  };
  constexpr int XRL = NT1;
  float px[2][XRL], pact[2] = {0.f, 0.f}, psc[2] = {0.f, 0.f};
  const float* const p_sc = gpart == 0 ? p_s0 : (gpart == 1 ? p_s1 : p_s2);
  const int sc_dst = opq(gpart == 0 ? S::OLP + gb_row : (gpart == 1 ? S::ADR + gb_row : (gpart == 2 ? S::ADC + gb_row : S::MISC + 46)));      // (gpart >= 3: a scratch word; + 16 for the second tile)
  const int o_acs = opq(S::ACT + gb_row * SAX + gpart);
  auto issue_rows = [&](int t, int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;
    const unsigned ob = off * (unsigned)O;
#pragma unroll
    for (int i = 0; i < XRL; ++i) { const int k = gpart + 16 * i; px[t][i] = p_obs[ob + (unsigned)(k < O ? k : O - 1)]; }
    pact[t] = p_act[off * (unsigned)AS + (unsigned)(gpart < AS ? gpart : AS - 1)];      // (the critics fetch the action bytes too: a load is cheaper than a branch here)
    psc[t] = p_sc[off];
  };
  auto commit_rows = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < XRL; ++i) { const int k = gpart + 16 * i; sm[S::XT + k * STX + 16 * t + gb_row] = k < O ? px[t][i] : 0.f; }
      if (role == 0) sm[o_acs + 16 * t * SAX] = gpart < AS ? pact[t] : 0.f;
      sm[sc_dst + 16 * t] = psc[t];
    }
  };
  // advantage statistics of a minibatch (policy role): thread 64 + i, i < nb (<= 128: waves 1 and 2) holds row i's (A_r, A_c)
  float sar = 0.f, sac = 0.f;
  auto issue_stats = [&](int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;     // rows beyond the minibatch are masked in stats_partials
    sar = p_s1[off];
    sac = p_s2[off];
  };
  auto stats_partials = [&](int nb) {      // (all four parts: the statistics are the whole minibatch's, formed identically)
    if (role != 0 || w == 0 || w == 3) return;
    const bool in = stid < nb;
    const float v_r = in ? sar : 0.f, v_c = in ? sac : 0.f, v_rr = in ? sar * sar : 0.f;
    const float s_r = wave_sum_fast(v_r), s_c = wave_sum_fast(v_c), s_rr = wave_sum_fast(v_rr);
    if (lane == 0) { sm[o_msc + 3 * (w - 1)] = s_r; sm[o_msc + 3 * (w - 1) + 1] = s_c; sm[o_msc + 3 * (w - 1) + 2] = s_rr; }
  };
  float mean_r = 0.f, istd_r = 1.f, mean_c = 0.f;
  auto read_stats = [&](int nb) {
    if (role != 0) return;
    const float s_r = sm[o_msc + 0] + sm[o_msc + 3];
    const float s_c = sm[o_msc + 1] + sm[o_msc + 4];
    const float s_rr = sm[o_msc + 2] + sm[o_msc + 5];
    const float inv = __builtin_amdgcn_rcpf((float)nb);
    mean_r = s_r * inv;
    mean_c = s_c * inv;
    // unbiased variance from the raw moments (advantages are O(1): fp32 cancellation stays ~1e-6 relative)
    const float var = fmaxf(s_rr - s_r * mean_r, 0.f) * __builtin_amdgcn_rcpf((float)(nb - 1));
    istd_r = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var) + 1e-8f);
  };
  auto refresh_gauss = [&]() {   // wave 1, lanes q == 0 own log_std r: derived constants of the Gaussian head
    if (!DISC && role == 0 && w == 1 && q == 0) {
      const float wex = sm[ex_s];      // (= S::LS + r on these lanes)
      const float sd = __expf(wex);
      const float iv = __builtin_amdgcn_rcpf(sd * sd);
      sm[o_gau + r] = r < A ? iv : 0.f;          // (q == 0: o_gau = S::GAU)
      sm[o_gau + 16 + r] = r < A ? 0.5f * iv : 0.f;
      sm[o_gau + 32 + r] = r < A ? wex + LOG_SQRT_2PI_F : 0.f;     // log(sd) = log_std
      const float ent = row_sum(r < A ? HALF_LOG_2PI_PLUS_HALF_F + wex : 0.f);
      if (r == 0) sm[o_msc + 22] = ent;
    }
  };
  // ---- synchronisation inside the quad (ppo_train_halves.hip): a phase counter per wave in LDS
  int* const pflag = reinterpret_cast<int*>(sm + o_msc + 48);      // [4] one word per wave
  int pphase = 0;
  auto quad_signal = [&]() {
    if (ICRL_QW2_QUAD_BARRIER) return;      // (the barrier in quad_wait does both)
    ++pphase;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(pflag + w, pphase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto quad_wait = [&]() {
    if (ICRL_QW2_QUAD_BARRIER) { lds_barrier(); return; }
    while (true) {
      const int f0 = __hip_atomic_load(pflag + qp0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int f1 = __hip_atomic_load(pflag + qp1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int f2 = __hip_atomic_load(pflag + qp2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int f = f0 < f1 ? (f0 < f2 ? f0 : f2) : (f1 < f2 ? f1 : f2);
      if (f >= pphase) break;
      __builtin_amdgcn_s_sleep(0);
    }
    asm volatile("" ::: "memory");
  };
  // the running statistics live in lane 0 of wave 3 of each role
  const bool book = tid == 192;
  float st_ent = 0.f, st_pg = 0.f, st_vl = 0.f, st_cf = 0.f, last_loss = 0.f, kl_sum = 0.f;
  int steps_done = 0, early_stop_epoch = a.hp.n_epochs, status = 0;

  // ---- pipeline prologue: rows of step s + 1 are loaded during step s, their permutation indices a step earlier, the plan entries a step before that
  int idx_next[2], idx_nx2[2];
  int2 pc_nx3[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    idx_next[t] = chunk_idx(ld_chunk(2 + t));
    idx_nx2[t] = chunk_idx(ld_chunk(4 + t));
    pc_nx3[t] = ld_chunk(6 + t);
    issue_rows(t, chunk_idx(ld_chunk(t)));
  }
  int4 ps_next = ld_step(0), ps_nx2 = ld_step(1), ps_nx3 = ld_step(2);
  issue_stats(stat_idx(ps_next));
  int sidx_next = stat_idx(ps_nx2);
  if (tid == 0) sm[o_msc + 14] = run_on_one_xcd(xch0, slot_j, 12, true) ? 1.f : 0.f;
  __syncthreads();                      // initial weights visible (refresh_gauss reads log_std)
#ifdef ICRL_ASSUME_XCD_LOCAL      // (measurement only: what a compile-time store scope would buy — no branch per exchange store: 11.61 -> 11.48 us per step (-1.1 %): not worth a verify-and-relaunch protocol in the ABI)
  constexpr bool xcd_local = true;
#else
  const bool xcd_local = __builtin_amdgcn_readfirstlane(__float_as_int(sm[o_msc + 14])) != 0;
#endif
  refresh_gauss();
  commit_rows();
  stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  __syncthreads();
  read_stats(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  const float inv_n_mb = 1.f / (float)((T * N + a.hp.batch_size - 1) / a.hp.batch_size);

  const int b = r;                      // this lane's row of a tile (all four q lanes share it)
  float* const pt = sm + (4 * q) * STX + b;     // + image + (16 f + i) STX + 16 t: element [feature 16 f + 4 q + i][row 16 t + b]

  // ---- the exchange of the partial gradients (ppo_train_quarters.hip)
  typedef unsigned int raw_u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gxp, 0, (int)ICRL_PPO_SPLIT_BYTES, 0x00020000);
  constexpr int NGRP = NT1 + 6;                               // groups: W1 tiles, 4 W2 tiles, head, {b1, b2, extra, book-keeping sum 0}
  constexpr int XFLAG = (NGRP + 1) * THQ2 * 16;               // + the book-keeping lane's record, then one flag word per wave (64 B apart)
  constexpr int XBLK = XFLAG + 4 * 64;                        // bytes of one (parity, role, part) block
  static_assert(24 * XBLK + 4 * 512 <= (int)ICRL_PPO_SPLIT_BYTES, "the exchange of four row parts per network lives in the split workspace");
  auto raw_store = [&](int byte_off, const f32x4& v) {
    const raw_u4 u = __builtin_bit_cast(raw_u4, v);
    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(u, grs, byte_off, 0, 1);
    else __builtin_amdgcn_raw_buffer_store_b128(u, grs, byte_off, 0, 16);
  };
  auto raw_load = [&](int byte_off) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, byte_off, 0, 16)); };

  constexpr bool prof = PROF;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
    ps_nx3 = ld_step(st + 3 < n_steps + 2 ? st + 3 : n_steps + 1);     // (the table carries two zero entries at its end)
    const int nb = ps.nb_flags & NB_MASK;
    const float inv_nb = __builtin_amdgcn_rcpf((float)nb);
    const float c_mean_r = mean_r, c_mean_c = mean_c, c_istd_r = istd_r;   // statistics of THIS minibatch
    const float cpol_nb = inv_nb * __builtin_amdgcn_rcpf(1.f + nu);
    issue_stats(sidx_next);
    sidx_next = stat_idx(ps_nx2);
    float mb_s0 = 0.f, mb_s1 = 0.f, mb_s2 = 0.f, mb_s3 = 0.f, mb_s4 = 0.f;  // bookkeeping lane: minibatch sums of the loss statistics
    const int xrole = ((int)(step & 1) * 3 + role) * 4 * XBLK;      // the four blocks of this role and step parity
    const int xmine = xrole + part * XBLK;
    auto publish4 = [&](int g, const f32x4& v) {
      if (ICRL_QW2_EARLY_PUBLISH) raw_store(xmine + (g * THQ2 + tid) * 16, v);
    };
    // row tile t = rows 16 part .. of chunk t: chunk 0 holds min(nb, 64) rows, chunk 1 the rest
    bool valid[2];
    valid[0] = 16 * part + b < (nb < RB ? nb : RB);
    valid[1] = 16 * part + b < nb - RB;

    // ================= forward: feature tile w of every layer, both row tiles =================
    f32x4 h1c[2], h2c[2], outc[2];
    float pl_olp[2] = {0.f, 0.f}, pl_adr[2] = {0.f, 0.f}, pl_adc[2] = {0.f, 0.f};      // (ICRL_QW2_LOSS_PRELOAD)
    f32x4 pl_act[2], pl_iv, pl_hiv, pl_lsd, pl_wht;
    pl_act[0] = pl_act[1] = pl_iv = pl_hiv = pl_lsd = pl_wht = f32x4{0.f, 0.f, 0.f, 0.f};
    {  // layer 1: the weight operands once, two independent MFMA chains
      const float* pa = sm + o_w1a;
      f32x4 aw[NT1];
#pragma unroll
      for (int js = 0; js < NT1; ++js) aw[js] = lds128(pa + 16 * js);
      float bx[2][NT1][4];              // x[row 16 t + b][k = 16 js + 4 q + e]
      const float* pb = sm + S::XT + (4 * q) * STX + b;
#pragma unroll
      for (int js = 0; js < NT1; ++js)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 2; ++t) bx[t][js][e] = (OBS == 0 || 16 * js + e < OBS) ? pb[(16 * js + e) * STX + 16 * t] : 0.f;
      f32x4 z[2];
      z[0] = z[1] = lds128(sm + o_bia);      // the bias is the accumulator's initial value
      // every operand fetch is issued before the first MFMA (left to itself the compiler fetches one k step, waits for it, issues its two MFMAs:
      // the full LDS latency in front of each of the 29 steps)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int js = 0; js < NT1; ++js)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (OBS == 0 || 16 * js + e < OBS) {      // (k groups 16 js + 4 q + e all beyond obs: nothing to add)
            z[0] = MFMA_F32(aw[js][e], bx[0][js][e], z[0]);
            z[1] = MFMA_F32(aw[js][e], bx[1][js][e], z[1]);
          }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h1c[t][i] = fast_tanh(z[t][i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::H1T + (16 * w + i) * STX + 16 * t] = h1c[t][i];
        *reinterpret_cast<f32x4*>(sm + S::H1R + (16 * t + b) * SRX + 16 * w + 4 * q) = h1c[t];
      }
    }
    {  // prefetch the next step's rows (random pieces of the rollout buffer: several microseconds away)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int idx_now = idx_next[t];
        idx_next[t] = idx_nx2[t];
        idx_nx2[t] = chunk_idx(pc_nx3[t]);
        pc_nx3[t] = ld_chunk(2 * (st + 4) + t < 2 * n_steps + 8 ? 2 * (st + 4) + t : 2 * n_steps + 8);      // (ten zero entries behind the table)
        issue_rows(t, idx_now);
      }
    }
    quad_signal();               // (P1) this wave's features of h1 are complete
    {  // layer 2: own quarter of K from registers, the other three from the row-major image
      const float* pa = sm + o_w2a;
      const f32x4 awo = lds128(pa + 16 * w);
      f32x4 awp[3];
#pragma unroll
      for (int d = 1; d < 4; ++d) awp[d - 1] = lds128(pa + 16 * ((w + d) & 3));
      f32x4 z[2];
      z[0] = z[1] = lds128(sm + o_bia + HD);
#pragma unroll
      for (int e = 0; e < 4; ++e) { z[0] = MFMA_F32(awo[e], h1c[0][e], z[0]); z[1] = MFMA_F32(awo[e], h1c[1][e], z[1]); }
      quad_wait();               // the other three waves' features of h1 are complete
      f32x4 hp[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d = 1; d < 4; ++d) hp[t][d - 1] = lds128(sm + S::H1R + (16 * t + b) * SRX + 4 * q + 16 * ((w + d) & 3));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) { z[0] = MFMA_F32(awp[d][e], hp[0][d][e], z[0]); z[1] = MFMA_F32(awp[d][e], hp[1][d][e], z[1]); }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h2c[t][i] = fast_tanh(z[t][i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::H2T + (16 * w + i) * STX + 16 * t] = h2c[t][i];
      }
    }
    {  // head, split over K four ways; partial tiles exchanged through LDS, summed as (p0 + p1) + (p2 + p3) by all four waves
      const f32x4 aw = lds128(sm + o_wha);
      f32x4 acc[2];
      acc[0] = acc[1] = w == 0 ? lds128(sm + o_bh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[0] = MFMA_F32(aw[e], h2c[0][e], acc[0]); acc[1] = MFMA_F32(aw[e], h2c[1][e], acc[1]); }
      float* const hpx = sm + o_hpx;
      *reinterpret_cast<f32x4*>(hpx + w * 256) = acc[0];
      *reinterpret_cast<f32x4*>(hpx + 1024 + w * 256) = acc[1];
      if (ICRL_QW2_LOSS_PRELOAD) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          pl_olp[t] = sm[o_sid + 16 * t]; pl_adr[t] = sm[o_sid + 32 + 16 * t]; pl_adc[t] = sm[o_sid + 64 + 16 * t];
          if (role == 0 && !DISC) pl_act[t] = lds128(sm + o_act + 16 * t * SAX);
        }
        if (role == 0 && !DISC) { pl_iv = lds128(sm + o_gau); pl_hiv = lds128(sm + o_gau + 16); pl_lsd = lds128(sm + o_gau + 32); }
        pl_wht = lds128(sm + o_wht);
      }
      quad_signal(); quad_wait();  // (P3) all four partial tiles of both row tiles stored
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 p0 = lds128(hpx + 1024 * t), p1 = lds128(hpx + 1024 * t + 256), p2 = lds128(hpx + 1024 * t + 512), p3 = lds128(hpx + 1024 * t + 768);
#pragma unroll
        for (int i = 0; i < 4; ++i) outc[t][i] = (p0[i] + p1[i]) + (p2[i] + p3[i]);
      }
    }
    STAMP(0)   // forward
    // ============ loss + d loss / d head output, in the C layout (all four waves: identical values), both row tiles ============
    f32x4 dout[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      dout[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;   // per-row statistics (identical in the four q lanes)
      if (role == 0) {
        float lp = 0.f, ent = 0.f;
        f32x4 g1 = f32x4{0.f, 0.f, 0.f, 0.f}, g2 = f32x4{0.f, 0.f, 0.f, 0.f};   // d log-prob / d out, second term
        if (DISC) {
          // Categorical(logits) (ref: distributions.py:274-288)
          float lg[4], zmax = -INFINITY;
#pragma unroll
          for (int i = 0; i < 4; ++i) { lg[i] = (4 * q + i < A) ? outc[t][i] : -INFINITY; zmax = fmaxf(zmax, lg[i]); }
          zmax = xor16_max(xor32_max(zmax));
          float se = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) se += (4 * q + i < A) ? expf(lg[i] - zmax) : 0.f;
          se = quad_rows_sum(se);
          const float lse = zmax + logf(se);
          const int act = (int)sm[o_act - 4 * q + 16 * t * SAX];
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
          const f32x4 actv = ICRL_QW2_LOSS_PRELOAD ? pl_act[t] : lds128(sm + o_act + 16 * t * SAX);
          const f32x4 iv = ICRL_QW2_LOSS_PRELOAD ? pl_iv : lds128(sm + o_gau), hiv = ICRL_QW2_LOSS_PRELOAD ? pl_hiv : lds128(sm + o_gau + 16),
                      lsd = ICRL_QW2_LOSS_PRELOAD ? pl_lsd : lds128(sm + o_gau + 32);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float dd = actv[i] - outc[t][i];                 // pad actions / outputs are 0
            lp += -(dd * dd) * hiv[i] - lsd[i];
            g1[i] = dd * iv[i];
            g2[i] = (4 * q + i < A) ? (dd * dd) * iv[i] - 1.f : 0.f;   // d log-prob / d log_std
          }
          lp = quad_rows_sum(lp);
        }
        const float old_lp = ICRL_QW2_LOSS_PRELOAD ? pl_olp[t] : sm[o_sid + 16 * t];
        const float ratio = __expf(lp - old_lp);
        const float Ar = ((ICRL_QW2_LOSS_PRELOAD ? pl_adr[t] : sm[o_sid + 32 + 16 * t]) - c_mean_r) * c_istd_r;
        const float Ac = (ICRL_QW2_LOSS_PRELOAD ? pl_adc[t] : sm[o_sid + 64 + 16 * t]) - c_mean_c;
        const float s1 = Ar * ratio;
        const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const float s2 = Ar * rc;
        const float gsel = (s1 <= s2) ? Ar : 0.f;                       // d min(s1, s2) / d ratio
        const float dlp = valid[t] ? cpol_nb * (-gsel + nu * Ac) * ratio : 0.f;  // d loss / d log_prob
        if (DISC) {
          const float dent = valid[t] ? ent_coef * inv_nb : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) dout[t][i] = dlp * g1[i] + dent * g2[i];
        } else {
          f32x4 tt;      // d log_std: sum over this tile's 16 rows, per output o = 4 q + i
#pragma unroll
          for (int i = 0; i < 4; ++i) { dout[t][i] = dlp * g1[i]; tt[i] = row_sum(dlp * g2[i]); }
          if (w == 0 && r == 0) *reinterpret_cast<f32x4*>(sm + o_msc + (S::PLS - S::MISC) + 16 * t + 4 * q) = tt;
        }
        const bool cnt = valid[t] && q == 0;
        v0 = cnt ? fminf(s1, s2) : 0.f; v1 = cnt ? Ac * ratio : 0.f; v2 = (cnt && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
        v3 = cnt ? old_lp - lp : 0.f; v4 = cnt ? ent : 0.f;
      } else {
        const float v = quad_rows_sum(q == 0 ? outc[t][0] : 0.f);      // lane (r, q = 0) holds output 0 of row b
        const float R = ICRL_QW2_LOSS_PRELOAD ? pl_adr[t] : sm[o_sid + 32 + 16 * t];
        float vp = v, pass = 1.f;
        if (vclip >= 0.f) {
          const float old = ICRL_QW2_LOSS_PRELOAD ? pl_olp[t] : sm[o_sid + 16 * t];
          const float dv = v - old;
          vp = old + fminf(fmaxf(dv, -vclip), vclip);
          pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
        }
        const float e = vp - R;
        const float d0 = valid[t] ? vcoef * 2.f * e * inv_nb * pass : 0.f;
        dout[t][0] = q == 0 ? d0 : 0.f;
        v0 = (valid[t] && q == 0) ? e * e : 0.f;
      }
      if (w == 0) {     // one wave of the quad reports the tile's statistics
        v0 = row_sum(v0); v1 = row_sum(v1); v2 = row_sum(v2); v3 = row_sum(v3);
        if (DISC) v4 = row_sum(v4);
        if (lane == 0) { float* pst = sm + o_msc + (S::PST - S::MISC) + 8 * t; pst[0] = v0; pst[1] = v1; pst[2] = v2; pst[3] = v3; pst[4] = v4; }
      }
    }
    STAMP(1)   // loss
    // ================= backward of the activations =================
    f32x4 dz2c[2], dz1c[2];
    {  // dH2^T = Wh^T . dOut^T for the own feature tile: A = WHT[j = 16 w + r][o = 4 q + e] (K = 16 outputs)
      const f32x4 aw = ICRL_QW2_LOSS_PRELOAD ? pl_wht : lds128(sm + o_wht);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[0] = MFMA_F32(aw[e], dout[0][e], acc[0]); acc[1] = MFMA_F32(aw[e], dout[1][e], acc[1]); }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dz2c[t][i] = fmaf(-(h2c[t][i] * h2c[t][i]), acc[t][i], acc[t][i]);   // acc (1 - h2^2)
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::DZ2T + (16 * w + i) * STX + 16 * t] = dz2c[t][i];
        *reinterpret_cast<f32x4*>(sm + S::DZ2R + (16 * t + b) * SRX + 16 * w + 4 * q) = dz2c[t];
        if (w == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::DOT + i * STX + 16 * t] = dout[t][i];
        }
      }
    }
    quad_signal();               // (P4) this wave's features of dz2 complete
    {  // dH1^T = W2^T . dz2^T: A = W2T[k = 16 w + r][j = 16 js + 4 q + e]; own quarter of K before the wait for the others
      const float* pa = sm + o_w2a + (S::W2T - S::W2);
      const f32x4 awo = lds128(pa + 16 * w);
      f32x4 awp[3];
#pragma unroll
      for (int d = 1; d < 4; ++d) awp[d - 1] = lds128(pa + 16 * ((w + d) & 3));
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[0] = MFMA_F32(awo[e], dz2c[0][e], acc[0]); acc[1] = MFMA_F32(awo[e], dz2c[1][e], acc[1]); }
      quad_wait();               // the other three waves' features of dz2 complete
      f32x4 dp[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d = 1; d < 4; ++d) dp[t][d - 1] = lds128(sm + S::DZ2R + (16 * t + b) * SRX + 4 * q + 16 * ((w + d) & 3));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[0] = MFMA_F32(awp[d][e], dp[0][d][e], acc[0]); acc[1] = MFMA_F32(awp[d][e], dp[1][d][e], acc[1]); }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dz1c[t][i] = fmaf(-(h1c[t][i] * h1c[t][i]), acc[t][i], acc[t][i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::DZ1T + (16 * w + i) * STX + 16 * t] = dz1c[t][i];
      }
    }
    STAMP(2)   // activation backward
    lds_barrier();  // (S5) the tiles' columns of h1^T, h2^T, dz1^T, dz2^T, dOut^T (and the loss partials) are complete
    // ================= weight gradients: rows 16 w .. of dW2 / dW1, columns 16 w .. of dWh; K = this part's 32 rows =================
    {
      f32x4 az[2];   // dz2^T[j = 16 w + r][rows 16 t + 4 q + e]
#pragma unroll
      for (int t = 0; t < 2; ++t) az[t] = lds128(sm + S::DZ2T + (16 * w + r) * STX + 16 * t + 4 * q);
      const float* pb = sm + S::H1T + r * STX + 4 * q;
      f32x4 bh[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) bh[c][t] = lds128(pb + c * 16 * STX + 16 * t);
      __builtin_amdgcn_sched_barrier(0);
      // two output tiles at a time: their MFMA chains (8 dependent steps each) interleave
#pragma unroll
      for (int c = 0; c < 4; c += 2) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc0 = MFMA_F32(az[t][e], bh[c][t][e], acc0); acc1 = MFMA_F32(az[t][e], bh[c + 1][t][e], acc1); }
        gW2r[c] = acc0; gW2r[c + 1] = acc1;
        if (c > 0) { publish4(NT1 + c - 2, gW2r[c - 2]); publish4(NT1 + c - 1, gW2r[c - 1]); }
      }
      const float s = ((az[0][0] + az[0][1]) + (az[0][2] + az[0][3])) + ((az[1][0] + az[1][1]) + (az[1][2] + az[1][3]));     // d b2[16 w + r]
      gb2r = quad_rows_sum(s);
    }
    {
      f32x4 ao[2], bh[2];   // dOut^T[o = r][rows], h2^T[j = 16 w + r][rows]
#pragma unroll
      for (int t = 0; t < 2; ++t) { ao[t] = lds128(sm + S::DOT + r * STX + 16 * t + 4 * q); bh[t] = lds128(sm + S::H2T + (16 * w + r) * STX + 16 * t + 4 * q); }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = MFMA_F32(ao[t][e], bh[t][e], acc[t]);
      publish4(NT1 + 2, gW2r[2]); publish4(NT1 + 3, gW2r[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) gWhr[i] = acc[0][i] + acc[1][i];
      float s = ((ao[0][0] + ao[0][1]) + (ao[0][2] + ao[0][3])) + ((ao[1][0] + ao[1][1]) + (ao[1][2] + ao[1][3]));      // head bias (wave 0 keeps it)
      s = quad_rows_sum(s);
      const float sl = sm[o_msc + (S::PLS - S::MISC) + r] + sm[o_msc + (S::PLS - S::MISC) + 16 + r];       // wave 1 (Gaussian policy): d log_std r = the two row tiles' partials
      gex = w == 0 ? s : ((!DISC && role == 0 && w == 1) ? sl : 0.f);
    }
    {
      f32x4 az[2];   // dz1^T[j = 16 w + r][rows]
#pragma unroll
      for (int t = 0; t < 2; ++t) az[t] = lds128(sm + S::DZ1T + (16 * w + r) * STX + 16 * t + 4 * q);
      const float* pb = sm + S::XT + r * STX + 4 * q;     // x^T[k = 16 c + r][rows 16 t + 4 q + e]
      f32x4 bx[NT1][2];
#pragma unroll
      for (int c = 0; c < NT1; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) bx[c][t] = lds128(pb + c * 16 * STX + 16 * t);
      __builtin_amdgcn_sched_barrier(0);
      static_assert(NT1 % 2 == 0, "observation tiles are walked in pairs");
#pragma unroll
      for (int c = 0; c < NT1; c += 2) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc0 = MFMA_F32(az[t][e], bx[c][t][e], acc0); acc1 = MFMA_F32(az[t][e], bx[c + 1][t][e], acc1); }
        gW1r[c] = acc0; gW1r[c + 1] = acc1;
        if (c == 0) publish4(NT1 + 4, gWhr); else { publish4(c - 2, gW1r[c - 2]); publish4(c - 1, gW1r[c - 1]); }
      }
      const float s = ((az[0][0] + az[0][1]) + (az[0][2] + az[0][3])) + ((az[1][0] + az[1][1]) + (az[1][2] + az[1][3]));
      gb1r = quad_rows_sum(s);
      publish4(NT1 - 2, gW1r[NT1 - 2]); publish4(NT1 - 1, gW1r[NT1 - 1]);
    }
    if (book) {
      const float* pst = sm + o_msc + (S::PST - S::MISC);
      mb_s0 = pst[0] + pst[8]; mb_s1 = pst[1] + pst[9]; mb_s2 = pst[2] + pst[10]; mb_s3 = pst[3] + pst[11];
      if (DISC) mb_s4 = pst[4] + pst[12];
    }
    STAMP(3)   // weight gradients

    // ================= partial gradients of this part <-> the other three parts of the same network =================
    {
      f32x4 gsc = f32x4{gb1r, gb2r, gex, book ? mb_s0 : 0.f};      // (the book-keeping lane's first loss sum rides in the spare slot)
      auto grp = [&](int g) -> f32x4& { return g < NT1 ? gW1r[g] : (g < NT1 + 4 ? gW2r[g - NT1] : (g == NT1 + 4 ? gWhr : gsc)); };
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        if (ICRL_QW2_EARLY_PUBLISH && g < NT1 + 5) continue;      // (already out, group by group, behind their GEMMs)
        raw_store(xmine + (g * THQ2 + tid) * 16, grp(g));
      }
      f32x4 bks = f32x4{mb_s1, mb_s2, mb_s3, mb_s4};
      if (book) raw_store(xmine + NGRP * THQ2 * 16, bks);
      // every store of this wave has been acknowledged (it is in the L2 the peers read through, or beyond) -> the wave's flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        if (xcd_local) __builtin_amdgcn_raw_buffer_store_b32(step, grs, xmine + XFLAG + 64 * w, 0, 1);
        else __builtin_amdgcn_raw_buffer_store_b32(step, grs, xmine + XFLAG + 64 * w, 0, 16);
      }
      stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);      // (the next minibatch's advantage statistics, inside the hop)
      // (own + partner) + (the other pair): the same floats in the same association on all four parts
      const int xa = xrole + (part ^ 1) * XBLK, xb = xrole + (part ^ 2) * XBLK, xc = xrole + (part ^ 3) * XBLK;
      bool timed_out = false;
      {   // the flags of this wave's three peers in one look (three loads in flight)
        const int fo = XFLAG + 64 * w;
        int spins = 0;
        while (true) {
          const unsigned fa = __builtin_amdgcn_raw_buffer_load_b32(grs, xa + fo, 0, 16);
          const unsigned fb = __builtin_amdgcn_raw_buffer_load_b32(grs, xb + fo, 0, 16);
          const unsigned fc = __builtin_amdgcn_raw_buffer_load_b32(grs, xc + fo, 0, 16);
          if (((fa ^ step) | (fb ^ step) | (fc ^ step)) == 0u) break;      // (no short-circuit: `fa == step && ...` lets the compiler sink the second and third look behind the first compare)
          if (++spins >= (1 << 22)) { timed_out = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
      }
      constexpr int DEP = ICRL_QW2_DEP;
      f32x4 ra[DEP], rb[DEP], rc[DEP];
#pragma unroll
      for (int k = 0; k < DEP; ++k)
        if (k < NGRP) {
          const int o = (k * THQ2 + tid) * 16;
          ra[k] = raw_load(xa + o); rb[k] = raw_load(xb + o); rc[k] = raw_load(xc + o);
        }
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        f32x4& v = grp(g);
        const f32x4 ca = ra[g % DEP], cb = rb[g % DEP], cc = rc[g % DEP];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (v[i] + ca[i]) + (cb[i] + cc[i]);
        if (g + DEP < NGRP) {
          const int o = ((g + DEP) * THQ2 + tid) * 16;
          ra[g % DEP] = raw_load(xa + o); rb[g % DEP] = raw_load(xb + o); rc[g % DEP] = raw_load(xc + o);
        }
      }
      gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
      if (book) {
        mb_s0 = gsc[3];
        const int o = NGRP * THQ2 * 16;
        const f32x4 ca = raw_load(xa + o), cb = raw_load(xb + o), cc = raw_load(xc + o);
        mb_s1 = (bks[0] + ca[0]) + (cb[0] + cc[0]); mb_s2 = (bks[1] + ca[1]) + (cb[1] + cc[1]);
        mb_s3 = (bks[2] + ca[2]) + (cb[2] + cc[2]); mb_s4 = (bks[3] + ca[3]) + (cb[3] + cc[3]);
      }
      if (timed_out) sm[o_msc + 13] = 1.f;       // reported through the status word like a timed-out norm exchange
    }

    // entropy term of the Gaussian policy loss: d(ent_coef * -mean(H)) / d log_std = -ent_coef (once, on the summed gradient)
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
      const u64 gran = ((u64)tag << 32) | (u64)__float_as_uint(ss);
      if (xcd_local) __hip_atomic_store(nx + (step & 1) * 32 + role * 8 + w, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(nx + (step & 1) * 32 + role * 8 + w, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // this workgroup reads its OWN four partials from LDS (same floats, same summation order as everybody else's view of them)
      sm[o_msc + 24 + role * 8 + w] = ss;
      if (book && role == 0) sm[o_msc + 12] = want_stop ? 1.f : 0.f;
      if (book) {
        ++steps_done;
        if (role == 0) {
          float ent = 0.f;
          if (DISC) ent = mb_s4 * inv_nb;
          else ent = sm[o_msc + 22];
          const float entropy_loss = -ent;
          const float pl = (-(mb_s0 * inv_nb) + nu * (mb_s1 * inv_nb)) * __builtin_amdgcn_rcpf(1.f + nu);
          st_ent += entropy_loss; st_pg += pl; st_cf += mb_s2 * inv_nb;
          last_loss = pl + ent_coef * entropy_loss;
          if (last_mb && part == 0) { float* stats = KARGS()->stats; stats[32 + epoch] = mean_kl; stats[7] = mean_kl; }
        } else {
          const float vl = mb_s0 * inv_nb;
          st_vl += vl;
          last_loss = vl;
        }
      }
    }
    STAMP(4)   // exchange + gradient norm + publish
    const int nb_next = __builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK;
    if (tid < 24 && (tid & 7) < 4 && (tid >> 3) != role) {      // the other two networks' granules of the same part
      u64 v = 0;
      int spins = 0;
      bool ok = false;
      const u64* const slot = nx + (step & 1) * 32 + tid;
      while (spins < (1 << 24)) {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)((v >> 32) & 0x7fffffffu) == step) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
        ++spins;
      }
      sm[o_msc + 24 + tid] = __uint_as_float((unsigned)(v & 0xffffffffu));
      if (tid == 3) sm[o_msc + 12] = (v >> 63) ? 1.f : 0.f;     // the granule of the policy's book-keeping wave carries the stop flag
      if (!ok) sm[o_msc + 13] = 1.f;
    }
    lds_barrier();   // (S6) norm partials and the next minibatch's statistics visible; nobody reads X^T / the per-row side data any more
    STAMP(5)   // granule wait
    commit_rows();   // the next minibatch (single X^T buffer: every wave is past its dW1) — visible behind (S7)
    float total = 0.f;
    {
#pragma unroll
      for (int g = 0; g < 3; ++g) {     // fixed order: every wave of every role and part forms the same total
        const f32x4 n = lds128(sm + o_msc + 24 + 8 * g);
        total += (n[0] + n[1]) + (n[2] + n[3]);
      }
      const f32x4 fl = lds128(sm + o_msc + 12);
      stop = fl[0] != 0.f;
      if (fl[1] != 0.f) { status = 1; stop = true; }
    }
    total = __builtin_amdgcn_sqrtf(total);
    float coef = max_grad_norm * __builtin_amdgcn_rcpf(total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    read_stats(nb_next > 0 ? nb_next : 2);

    // ================= Adam (torch.optim.Adam, single-tensor form) on the wave's own elements (ppo_train_rows.hip) =================
    {
      const float step_size = ps.step_size, inv_bc2_sqrt = ps.inv_bc2_sqrt;
      const float epsf = adam_epsf;
      const float omw1 = 1.f - w1, b2f_ = adam_b2f;
      const float cw1 = coef * w1, c2w2 = (coef * coef) * w2;
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
        f32x4 g_ = f32x4{gb1r, gb2r, ex_g >= 0 ? gex : 0.f, 0.f}, p_ = f32x4{sm[o_bo], sm[o_bo + HD], sm[ex_s], 0.f};
        float m3[4], v3[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) { m3[i] = mA[E_B1 + i]; v3[i] = vA[E_B1 + i]; }
        acc_put(m3[3], 0.f); acc_put(v3[3], 0.f);
        adam4(g_, m3, v3, p_);
#pragma unroll
        for (int i = 0; i < 3; ++i) { mA[E_B1 + i] = m3[i]; vA[E_B1 + i] = v3[i]; }
        if (q == 0) { sm[o_bo] = p_[0]; sm[o_bo + HD] = p_[1]; sm[ex_s] = p_[2]; }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this lane's log_std store has landed before refresh_gauss re-reads it
      refresh_gauss();
    }
    lds_barrier();   // (S7) updated weights and the next minibatch's rows visible
    STAMP(6)   // Adam
  }  // optimiser steps

  __syncthreads();
  // ---- write back weights, moments, statistics: the parts are replicas, part 0 writes
  if (part == 0) {
  const TrainArgs* kw = ka;
  asm volatile("" : "+s"(kw));
  const TrainArgs& a = *kw;
  const PolLayout& L = a.L;
  const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
  const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
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
    a.params[gb1 + jb] = sm[o_bo]; a.exp_avg[gb1 + jb] = acc_get(mA[E_B1]); a.exp_avg_sq[gb1 + jb] = acc_get(vA[E_B1]);
    a.params[gb2 + jb] = sm[o_bo + HD]; a.exp_avg[gb2 + jb] = acc_get(mA[E_B2]); a.exp_avg_sq[gb2 + jb] = acc_get(vA[E_B2]);
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

template <int NT1, bool DISC, int OBS, bool PROF>
__global__ void __launch_bounds__(THQ2) ppo_train_quarters2_kernel(TrainArgs a, int packed) {
  int run = 0, j = (int)blockIdx.x;
  if (packed && !packed_slot(12, 1, run, j)) return;
  ppo_train_quarters2_body<NT1, DISC, OBS, PROF>(a, (const TrainArgs*)__builtin_amdgcn_kernarg_segment_ptr(), j);
}

template <int NT1, bool DISC, int OBS, bool PROF>
static int launch_quarters2_p(const TrainArgs& a, hipStream_t s) {
  static_assert(SmemQ2<NT1>::TOTAL * sizeof(float) <= 160 * 1024, "LDS budget");
  const size_t bytes = ICRL_QW_STATIC_LDS ? 0 : (size_t)SmemQ2<NT1>::TOTAL * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)ppo_train_quarters2_kernel<NT1, DISC, OBS, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  TrainArgs arg = a;
  return launch_update_single(ppo_train_quarters2_kernel<NT1, DISC, OBS, PROF>, 12, dim3(THQ2), bytes, s, arg);
}
template <int NT1, bool DISC, int OBS>
static int launch_quarters2(const TrainArgs& a, hipStream_t s) {
  return (a.hp._pad & 1) ? launch_quarters2_p<NT1, DISC, OBS, true>(a, s) : launch_quarters2_p<NT1, DISC, OBS, false>(a, s);
}

// obs 65..128, minibatches of 65..128 rows, a.gx set and zeroed (prepare_train), the chunk plan with TWO entries per step; single-run launches
int launch_train_quarters_wide2(const TrainArgs& a, bool discrete, hipStream_t s) {
  if (a.L.O <= 64 || a.L.O > 128) return fail("update (four workgroups per network, two row tiles per wave): obs_dim %d outside 65..128", a.L.O);
  if (a.hp.batch_size <= RB || a.hp.batch_size > 2 * RB) return fail("update (four workgroups per network, two row tiles per wave): batch_size %d outside 65..128", a.hp.batch_size);
  if (!discrete && a.L.O == 113) return launch_quarters2<8, false, 113>(a, s);      // AntWall / AntWallBroken (BASELINE configs[2], [4])
  return discrete ? launch_quarters2<8, true, 0>(a, s) : launch_quarters2<8, false, 0>(a, s);
}

}  // namespace icrl
