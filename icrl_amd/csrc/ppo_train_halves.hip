// PPO-Lagrangian update, TWO workgroups per network, wave quads (obs_dim <= 32: HCWithPos, LapGridWorld) — gfx950.
//
// ref: stable_baselines3/ppo_lag/ppo_lag.py:196-299, common/buffers.py:594-627, common/policies.py:752-767,
//      common/distributions.py:143-171,274-288, torch.optim.Adam, clip_grad_norm_  (same contract as ppo_train_pairs.hip).
//
// The wave-pair kernel (ppo_train_pairs.hip) runs one network's 64-row chunk on ONE compute unit: 2 x 141 fp32 MFMAs per SIMD and
// optimiser step = 9.0 k cycles of a 19.9 k-cycle step during which nothing else issues on that SIMD (the fp32 MFMA runs on the
// SIMD's one fp32 lane array: co-execution counter 0).  Here a network's chunk is split by ROWS over two workgroups (grid 6 = 3
// networks x 2 halves, all on one XCD): workgroup (role, half) runs forward, loss, activation backward and the weight-gradient GEMMs
// of chunk rows 32 half .. 32 half + 31 — half the MFMAs, half the tanh / loss-tail work — and the two halves then exchange their
// partial gradients as RAW 16-byte stores + one flag word per wave (the protocol ppo_train_rows.hip's split mode measured at AntWall
// widths: 19.6 -> 18.1 us per step against tagged 8-byte granules), form own + partner — float addition is commutative bit for bit,
// so both halves hold identical gradients, run the identical norm / Adam arithmetic on identical weights and stay replicas.
//
// Inside a workgroup (8 waves, two per SIMD):
//   forward / activation backward: the 32 rows are two 16-row tiles; the FOUR waves of a row tile (rt2 = w & 1, fq = w >> 1) each own
//     ONE 16-feature tile of every layer.  Transposed GEMMs as in the pair kernel (Z^T = W . H^T: weights = A operand from LDS by
//     ds_read_b128, activations = B operand); a wave's own quarter of K comes from its registers, the other three quarters from a
//     ROW-major image [row][feature] the four waves write beside the [feature][row] image of the weight-gradient GEMMs (one
//     ds_write_b128 more per tile; the B operand of the partner tiles is then one ds_read_b128 instead of four ds_read_b32).  The
//     16-output head is split over K four ways (4 MFMAs each), the partial tiles summed in a fixed order by all four waves, which
//     then evaluate the loss tail on identical values.
//   weight gradients (K = the 32 rows): wave (jt = w & 3, kh = w >> 2) as in the pair kernel — rows 16 jt.. of dW2 for the column
//     tiles {2 kh, 2 kh + 1}, of dW1 for observation tile kh, (kh = 0) columns 16 jt.. of dWh — and it owns those elements' Adam state.
//   3 quad hand-offs (h1 | head partials | dz2) through LDS flags and 3 workgroup barriers per optimiser step.
//
// Built with -ffp-contract=off; FMA is used only where written (fmaf / MFMA).
#include <type_traits>

#include "ppo_common.h"

// (measured, round 5, us per optimiser step on one box: early publish + staging inside the hop 6.91 | early publish 6.89 | neither 6.87 |
// staging inside the hop alone 6.96 — all inside the run-to-run spread; the plain order ships)
#ifndef ICRL_HALVES_EARLY_PUBLISH
#define ICRL_HALVES_EARLY_PUBLISH 0
#endif
// the next minibatch is staged (rows -> the other X^T buffer, advantage statistics) between this wave's flag and its first look at the
// partner's: the exchange hop (~1.5 k cycles) is the longer of the step's two trips through the memory system and has no other
// independent work to run under; the norm granules of the other networks are then polled without anything in between
#ifndef ICRL_HALVES_STAGE_IN_HOP
#define ICRL_HALVES_STAGE_IN_HOP 0
#endif
// the first look at the other networks' norm granules is ISSUED before the staging of the next minibatch and read behind it: its trip to
// the L2 runs under the staging (6.84 -> 6.76 us per step, three alternating runs on one box; the pair kernel's same switch measured nothing
// in round 4, when the granules still crossed the fabric)
#ifndef ICRL_HALVES_EARLY_POLL
#define ICRL_HALVES_EARLY_POLL 1
#endif
// round 6 levers, all bit-identical to the plain order, all MEASURED SLOWER and off (tools/ab_train.sh, one box, us per
// optimiser step, two rounds of three launches: plain 6.74-6.78 | ADAM_TRIM 6.82-6.86 | ADAM_PRE 6.81-6.83 | LOSS_WAVES 2: 6.86-6.88 |
// LOSS_WAVES 1: 6.92-6.94 | ADAM_PRE + LOSS_WAVES 2: 6.94-6.97).  Fewer instructions do not shorten the step: the waves a lever relieves
// were not the ones the next hand-off waits for, and every lever adds a wave-uniform branch or a hand-off of its own.
// ICRL_HALVES_ADAM_TRIM: Adam only on the elements of the head / bias groups that can hold a parameter (head outputs are dealt four per
//   MFMA k group: 6 actions = 2 of a lane's 4 elements, a critic 1; the bias group {b1, b2, head bias | log_std, pad} = 2 or 3 of 4)
#ifndef ICRL_HALVES_ADAM_TRIM
#define ICRL_HALVES_ADAM_TRIM 0
#endif
// ICRL_HALVES_ADAM_PRE: the moments are scaled by beta1 / beta2 (m *= beta1, v *= beta2 — the part of Adam that needs neither the gradient
//   nor the clip coefficient) inside the first exchange hop, where the wave only waits for its partner's flag
#ifndef ICRL_HALVES_ADAM_PRE
#define ICRL_HALVES_ADAM_PRE 0
#endif
// ICRL_HALVES_LOSS_WAVES: how many of a quad's four waves evaluate the loss tail (4: all, on identical values; 2: one per SIMD of the quad;
//   1: one) — the others take d loss / d head output from an LDS record of the first
#ifndef ICRL_HALVES_LOSS_WAVES
#define ICRL_HALVES_LOSS_WAVES 4
#endif
// ICRL_HALVES_POLL_ROLL: four looks at the partner's flag in flight instead of one (see the exchange below): 6.86-6.89 against 6.77-6.79 us —
//   the extra looks queue in the L2 in front of the data; off
#ifndef ICRL_HALVES_POLL_ROLL
#define ICRL_HALVES_POLL_ROLL 0
#endif
// ICRL_HALVES_OWNER_ADAM (four parts only): a wave's parameters are summed, norm-ed and Adam-updated by ONE of the four parts — the OWNER of wave w
//   is part w / 2 — instead of by all four on replicas: only the owner fetches the three peers' partial gradients of that wave (a quarter of the
//   exchange's bytes per workgroup), Adam runs on one wave per SIMD, and the updated parameters travel back to the other three parts in a third hop
//   (the Adam moments live in the owner's registers only).  Same sums in the same order: the results are bit-identical to the replicated form.
#ifndef ICRL_HALVES_OWNER_ADAM
#define ICRL_HALVES_OWNER_ADAM 0
#endif
// ICRL_HALVES_FLAGS_TOGETHER (four parts): the three peers' flags polled in one look instead of one blocking look each (three trips through the L2 even
//   when all three are long set): 6.36-6.40 -> 6.25-6.30 us per step; on
#ifndef ICRL_HALVES_FLAGS_TOGETHER
#define ICRL_HALVES_FLAGS_TOGETHER 1
#endif
// ICRL_HALVES_PARTNER_SUM (four parts, late round 6): the four-way sum written as (own + partner) + (the other pair) on this wave's registers and three
//   straight-line fetches — part p's partner is p ^ 1, the other pair {p ^ 2, p ^ 3} — instead of add(fetch(0), fetch(1)) + add(fetch(2), fetch(3)) with a
//   run-time "is it my own block" branch and a struct copy per fetch (~1 000 instructions in the exchange phase).  Float addition is commutative bit for
//   bit, so every part still holds (q0 + q1) + (q2 + q3): results are bit-identical to the branchy form.
#ifndef ICRL_HALVES_PARTNER_SUM
#define ICRL_HALVES_PARTNER_SUM 1
#endif
// ICRL_HALVES_FENCES (late round 6): every LDS operand fetch of a GEMM is issued before its first MFMA (a scheduling fence between the fetches and the
//   MFMA stream): left to itself the compiler fetches one k step, waits for it and issues its MFMA — the full LDS latency in front of each step
#ifndef ICRL_HALVES_FENCES
#define ICRL_HALVES_FENCES 1
#endif
#define HFENCE() do { if (ICRL_HALVES_FENCES) __builtin_amdgcn_sched_barrier(0); } while (0)
// ICRL_HALVES_LOSS_PRELOAD (late round 6): the loss tail's per-row operands (actions, old log-prob / value, advantages / return, the Gaussian head's
//   constants) and the A operand of dH2 are fetched from LDS BEFORE the head hand-off (P3) instead of behind it: their latency runs under the wait
#ifndef ICRL_HALVES_LOSS_PRELOAD
#define ICRL_HALVES_LOSS_PRELOAD 1
#endif
// ICRL_HALVES_ADAM_PRELOAD (late round 6): a wave's own master weights (the LDS operand copies it updates) are fetched BEFORE the norm barrier (S6) instead
//   of at the top of Adam: nobody writes them between the two points, and their latency runs under the wait for the other networks' norm granules.
//   Measured SLOWER (6.09 against 5.98 us per step, three alternating rounds): off.  ICRL_HALVES_LOSS_PRELOAD: 6.00 -> 5.98, on.
#ifndef ICRL_HALVES_ADAM_PRELOAD
#define ICRL_HALVES_ADAM_PRELOAD 0
#endif
// ICRL_HALVES_LATE_PREFETCH (late round 6): a forward wave issues the next chunk's row loads (index pipeline + ~60 instructions of address arithmetic) BEHIND
//   its h1 hand-off (P1) and layer 2's operand fetches, under their latency, instead of in front of the hand-off the other three waves wait for
//   — measured 6.23 against 5.97 us per step: the gathered rows are what the exchange's `s_waitcnt vmcnt(0)` (store acknowledgement in front of the flag)
//   ends up waiting for, ~2.8 us after their issue; every cycle they start later is a cycle on the step.  Off.
#ifndef ICRL_HALVES_LATE_PREFETCH
#define ICRL_HALVES_LATE_PREFETCH 0
#endif
// ICRL_HALVES_PREFETCH_AT_ADAM (late round 6): the rows of the NEXT step's first chunk are requested at the top of Adam (right behind the norm barrier, when
//   the previous rows have been committed and no poll is in flight whose in-order return they could delay) and ONE STEP EARLIER — the rows a step's hop
//   stages were requested at the top of the previous step's Adam, ~5 us before the store drain in front of that hop has to wait for them, instead of
//   behind layer 1 of the same step, ~2.8 us before.  Minibatches of one chunk only (batch_size <= 64: HCWithPos, LapGridWorld; one register set).
//   Measured SLOWER as well (6.03 against 5.94 us per step, three alternating rounds; bit-identical results): the drain does not wait for the rows. Off.
#ifndef ICRL_HALVES_PREFETCH_AT_ADAM
#define ICRL_HALVES_PREFETCH_AT_ADAM 0
#endif
// ICRL_HALVES_ROLE_SPEC: see ROLE_T at ppo_train_halves_body — measured: the critics' step loops shrink from ~2 100 to ~1 500 instructions, the policy's
//   (the one the others wait for) to 1 984 with 6 scratch reloads: 6.03 against 6.01 us per step, nothing; off
#ifndef ICRL_HALVES_ROLE_SPEC
#define ICRL_HALVES_ROLE_SPEC 0
#endif
// ICRL_HALVES_IMAGES_FIRST: the LDS layout with the [feature][row] / [row][feature] images in front of the weights, as in ppo_train_quarters2.hip — at two
//   observation tiles the ISA of the step loop does not change (2 386 against 2 392 lines, no scratch traffic either way): not timed, off
#ifndef ICRL_HALVES_IMAGES_FIRST
#define ICRL_HALVES_IMAGES_FIRST 0
#endif
// ICRL_HALVES_STAGE_IDLE (four parts, late round 6): the row stream — index pipeline, ~60 instructions of address arithmetic and the gather loads per
//   step, the commit of the staged rows — runs on waves 4..7, which sit at the barrier during forward / loss / backward, instead of on waves 0..3,
//   where it stood between layer 1 and the h1 hand-off of every forward wave.  Measured SLOWER (5.98 against 5.81 us per step, three alternating rounds;
//   bit-identical): waves 4..7 then reach the exchange with the commit still to do.  Off.
#ifndef ICRL_HALVES_STAGE_IDLE
#define ICRL_HALVES_STAGE_IDLE 0
#endif
// ICRL_HALVES_SOFFSET: the exchange's loads / stores with the uniform part of the address in the scalar-offset operand (no vector add per access) — measured
//   SLOWER (5.94 against 5.85 us per step; the offsets cost scalar registers, which this kernel has none to spare); off.  The same in
//   ppo_train_quarters2.hip (with the row gather through one 64-bit base and immediate offsets): 11.70 against 11.66, nothing; not kept.
#ifndef ICRL_HALVES_SOFFSET
#define ICRL_HALVES_SOFFSET 0
#endif
// ICRL_HALVES_DEFER_STATS (late round 6): the reductions that only feed LOGGED sums and the d log_std partial (row sums over the tile, executed by the quad's
//   first wave) run BEHIND that wave's dz2 hand-off (P4) instead of in front of it — the other three waves wait for that hand-off.  Measured: 5.91 against
//   5.88 us per step (bit-identical): nothing; off
#ifndef ICRL_HALVES_DEFER_STATS
#define ICRL_HALVES_DEFER_STATS 0
#endif
// ICRL_HALVES_QUAD_BARRIER (four parts, late round 6): the quad's three hand-offs (h1 | head partials | dz2) as WORKGROUP barriers — waves 4..7, which take no
//   part in forward / loss / backward, simply execute the same three `s_barrier`s on their way to (S5); `s_waitcnt lgkmcnt(0); s_barrier` replaces a flag
//   store plus a polling loop of three LDS loads per look (ppo_train_quarters2.hip, where the quad is the whole workgroup: 11.68 -> 11.53 us)
//   — here 5.888 -> 5.872 us (three alternating rounds): forward + backward lose ~380 cycles, the wait for the other networks' norm granules gains ~300; on
#ifndef ICRL_HALVES_QUAD_BARRIER
#define ICRL_HALVES_QUAD_BARRIER 1
#endif
// ICRL_HALVES_POLL_WAVE (late round 6): which wave polls the other networks' norm granules.  Wave 0 carries the most work of the exchange (five gradient
//   groups to sum, 20 elements of the norm partial) and polled behind it; a kh = 1 wave (three groups) is through its own part several hundred cycles
//   earlier, so its look is in flight while the others finish: 5.880 -> 5.860 us per step (three alternating rounds; wave 7, the book-keeping wave: 5.99)
#ifndef ICRL_HALVES_POLL_WAVE
#define ICRL_HALVES_POLL_WAVE 4
#endif
// ICRL_HALVES_FIRST_LOOK (four parts): the first look at the peers' flags issued before the staging and read behind it: 6.20-6.25 against 6.21-6.26 (noise); off
#ifndef ICRL_HALVES_FIRST_LOOK
#define ICRL_HALVES_FIRST_LOOK 0
#endif
// ICRL_HALVES_NORM_LOOK_EARLY (four parts): the first look at the other networks' norm granules issued in front of the partial-gradient sums: 6.38-6.43 against
//   6.29-6.33 — the look queues in front of the partners' blocks in the L2 port that bounds the hop; off
#ifndef ICRL_HALVES_NORM_LOOK_EARLY
#define ICRL_HALVES_NORM_LOOK_EARLY 0
#endif
// A/B: the four waves of a quad on four SIMDs (rt2 = w >> 2) instead of two and two (rt2 = w & 1)
#ifndef ICRL_HALVES_QUAD_SPREAD
#define ICRL_HALVES_QUAD_SPREAD 0
#endif

namespace icrl {

constexpr int THH = 512;   // 8 waves, two per SIMD
constexpr int HR = 32;     // rows of a 64-row chunk one workgroup computes
constexpr int STH = 40;    // row stride of the [feature][row] images (32 rows + 8: conflict-free ds_read_b128)
constexpr int SRM = 72;    // row stride of the [row][feature] images
constexpr int SAH = 24;    // row stride of the per-row action block and of the transposed head weights
constexpr int HX_GROUPS = 7;                                   // exchange slots per thread: W1 tile, 2 W2 tiles, head, {b1, b2, extra}, 2 book-keeping records
constexpr int HX_FLAG = HX_GROUPS * THH * 16;                  // byte offset of the 8 flag words (64 B apart) of a block
constexpr int HX_BLK = HX_FLAG + 8 * 64;                       // bytes of one (parity, role, half) block
constexpr int H3_FLAG = 5 * THH * 16;                          // third hop (owner -> the other parts): 5 parameter groups per thread + 8 flag words per block
constexpr int H3_BLK = H3_FLAG + 8 * 64;
constexpr int H3_BASE = 24 * HX_BLK;
static_assert(H3_BASE + 24 * H3_BLK + 3 * 512 <= (int)ICRL_PPO_SPLIT_BYTES, "the exchange of up to four row parts per network lives in the split workspace");

template <int NT1>
struct SmemH {  // offsets in floats (multiples of 4)
  static constexpr int O16 = 16 * NT1, SX = O16 + 8;
#if ICRL_HALVES_IMAGES_FIRST
  // images first (ppo_train_quarters2.hip: a ds instruction reaches 64 KB beyond its address register)
  static constexpr int XT0 = 0;                // [16 NT1][STH] x^T of this half's rows: XT[k][row]
  static constexpr int XT1 = XT0 + O16 * STH;
  static constexpr int H1T = XT1 + O16 * STH;  // [64][STH] h1^T
  static constexpr int H2T = H1T + HD * STH;
  static constexpr int DZ1T = H2T + HD * STH;
  static constexpr int DZ2T = DZ1T + HD * STH;
  static constexpr int DOT = DZ2T + HD * STH;  // [16][STH] d loss / d head output, transposed
  static constexpr int H1R = DOT + 16 * STH;   // [32][SRM] h1, row-major
  static constexpr int DZ2R = H1R + HR * SRM;  // [32][SRM] dz2, row-major
  static constexpr int ACT = DZ2R + HR * SRM;  // [32][SAH] actions of this half's rows
  static constexpr int OLP = ACT + HR * SAH;   // [32] old log-prob | old value
  static constexpr int ADR = OLP + HR;         // [32] raw reward advantage | return
  static constexpr int ADC = ADR + HR;         // [32] raw cost advantage
  static constexpr int PST = ADC + HR;         // [2][8] per-row-tile loss statistics
  static constexpr int PLS = PST + 16;         // [2][16] per-row-tile d log_std partial sums
  static constexpr int MISC = PLS + 32;        // [64] granule values, flags, advantage-statistics partials
  static constexpr int GAU = MISC + 64;        // [3][16]
  static constexpr int B1 = GAU + 48;
  static constexpr int B2 = B1 + HD;
  static constexpr int BH = B2 + HD;
  static constexpr int LS = BH + 16;
  static constexpr int HPX = LS + 16;          // [2][4][64][4] head partial tiles of the quads
  static constexpr int DOX = HPX + 2048;       // [2][64][4]
  static constexpr int WH = DOX + 512;         // [16][SH]
  static constexpr int WHT = WH + 16 * SH;     // [64][SAH]
  static constexpr int W1 = WHT + HD * SAH;    // [64][SX]
  static constexpr int W2 = W1 + HD * SX;      // [64][SH]
  static constexpr int W2T = W2 + HD * SH;     // [64][SH]
  static constexpr int TOTAL = W2T + HD * SH;
#else
  static constexpr int W1 = 0;                 // [64][SX]
  static constexpr int W2 = W1 + HD * SX;      // [64][SH]
  static constexpr int W2T = W2 + HD * SH;     // [64][SH]  W2T[k][j] = W2[j][k]
  static constexpr int WH = W2T + HD * SH;     // [16][SH]
  static constexpr int WHT = WH + 16 * SH;     // [64][SAH]  WHT[j][o] = WH[o][j]
  static constexpr int B1 = WHT + HD * SAH;
  static constexpr int B2 = B1 + HD;
  static constexpr int BH = B2 + HD;
  static constexpr int LS = BH + 16;
  static constexpr int GAU = LS + 16;          // [3][16] per-action 1/var, 0.5/var, log(sd) + log(sqrt(2 pi))
  static constexpr int XT0 = GAU + 48;         // [16 NT1][STH] x^T of this half's rows: XT[k][row]
  static constexpr int XT1 = XT0 + O16 * STH;
  static constexpr int H1T = XT1 + O16 * STH;  // [64][STH] h1^T
  static constexpr int H2T = H1T + HD * STH;
  static constexpr int DZ1T = H2T + HD * STH;
  static constexpr int DZ2T = DZ1T + HD * STH;
  static constexpr int DOT = DZ2T + HD * STH;  // [16][STH] d loss / d head output, transposed
  static constexpr int H1R = DOT + 16 * STH;   // [32][SRM] h1, row-major (B operand of layer 2 for the other three waves of the quad)
  static constexpr int DZ2R = H1R + HR * SRM;  // [32][SRM] dz2, row-major (B operand of dH1)
  static constexpr int HPX = DZ2R + HR * SRM;  // [2][4][64][4] head partial tiles of the quads
  static constexpr int ACT = HPX + 2048;       // [32][SAH] actions of this half's rows
  static constexpr int OLP = ACT + HR * SAH;   // [32] old log-prob | old value
  static constexpr int ADR = OLP + HR;         // [32] raw reward advantage | return
  static constexpr int ADC = ADR + HR;         // [32] raw cost advantage
  static constexpr int PST = ADC + HR;         // [2][8] per-row-tile loss statistics
  static constexpr int PLS = PST + 16;         // [2][16] per-row-tile d log_std partial sums
  static constexpr int MISC = PLS + 32;        // [64] granule values, flags, advantage-statistics partials
  static constexpr int DOX = MISC + 64;        // [2][64][4] d loss / d head output of a quad's first wave, lane for lane (ICRL_HALVES_LOSS_WAVES < 4)
  static constexpr int TOTAL = DOX + 512;
#endif
};

#define KARGS() ([&]() { const TrainArgs* k_ = ka; asm volatile("" : "+s"(k_)); return k_; }())

// BATCH: the argument block was read from memory (batched launch): its pointers are marked as global-memory pointers (common.h:
// as_global; a pointer a kernel LOADS has no known address space and every access through it is a flat_load / flat_store)
// NQ = 2: two workgroups per network, 32 rows of every 64-row chunk each (round 5).  NQ = 4 (round 6): FOUR workgroups per network, 16 rows = ONE
// row tile each — the forward / loss / activation backward run on waves 0..3 alone, one per SIMD (the other four wait at the barrier), the
// weight-gradient GEMMs have K = 16, and the four partial gradients are summed in the fixed order (q0 + q1) + (q2 + q3) by all four.
template <int NT1>
__device__ __forceinline__ float* halves_smem() {
  __shared__ __attribute__((aligned(16))) float sm[SmemH<NT1>::TOTAL];
  return sm;
}

// ROLE_T >= 0 (late round 6, the single-run HCWithPos launch): the network this workgroup serves as a compile-time constant — the body is instantiated once
// per role and the kernel picks by blockIdx: every `role == 0` test, the role-dependent pointer / coefficient selects and the dead half of the loss
// tail fold away (fewer scalar registers live across the step loop, whose spills were ~100 v_readlane per step)
// PROF (late round 6): the diagnostic phase timers (hp._pad & 1) are a compile-time variant — as a run-time flag their eight 64-bit accumulators and the
// flag's lane mask sat in scalar registers across the step loop of EVERY launch (~100 instructions and 15 scalar reloads per step: 5.96 -> 5.81 us)
template <int NT1, bool DISC, int OBS, bool BATCH, int NQ, int ROLE_T = -1, bool PROF = false>
__device__ __forceinline__ void ppo_train_halves_body(const TrainArgs& a, const TrainArgs* const ka, const int slot_j) {
#define GPH(x) (BATCH ? as_global(x) : (x))
  using S = SmemH<NT1>;
  constexpr int SX = S::SX;
  static_assert(NT1 == 2, "one observation tile per weight-gradient wave half");
  float* const sm = halves_smem<NT1>();      // static: every image offset folds into an immediate (ppo_train_pairs.hip); ONE array for the per-role instantiations
  static_assert(NQ == 2 || NQ == 4, "two or four row parts per network");
  constexpr int HRQ = RB / NQ;   // rows of a 64-row chunk this workgroup computes
  constexpr int NJS = HRQ / 16;  // 16-row tiles among them
  // fault injection for the tests (hp._pad & 64): the last workgroup of the run leaves at once — every wait of the others is bounded, the launch ENDS
  // with the status word set and the host raises
  if ((a.hp._pad & 64) && slot_j == 3 * NQ - 1) return;
  const int role = ROLE_T >= 0 ? ROLE_T : slot_j % 3;   // 0 policy, 1 reward critic, 2 cost critic
  const int half = slot_j / 3;   // which HRQ rows of every 64-row chunk (the "part")
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // NQ == 4: the one quad on waves 0..3 (four SIMDs); NQ == 2: two quads, two SIMDs each (ICRL_HALVES_QUAD_SPREAD: A/B of the other mapping)
  constexpr bool SPREAD = NQ == 4 || ICRL_HALVES_QUAD_SPREAD;
  const int rt2 = SPREAD ? w >> 2 : w & 1, fq = SPREAD ? w & 3 : w >> 1;      // forward / activation backward: row tile, feature tile
  const int qp0 = SPREAD ? w ^ 1 : w ^ 2, qp1 = SPREAD ? w ^ 2 : w ^ 4, qp2 = SPREAD ? w ^ 3 : w ^ 6;      // the other three waves of the quad
  const bool fwd_wave = rt2 < NJS;         // (NQ == 4: waves 4..7 take no part in forward / loss / activation backward)
  constexpr bool OWNER = NQ == 4 && ICRL_HALVES_OWNER_ADAM;
  const bool owner = !OWNER || (w >> 1) == half;      // this part sums, norms and Adam-updates wave w's parameters (OWNER: one part per wave)
  // the next minibatch is staged INSIDE the first exchange hop when there are four parts: that hop then moves three partners' blocks through the
  // compute unit's L2 port and is long enough to hide the staging (6.43-6.47 -> 6.34-6.40 us per step; with two parts 6.96 against 6.87)
  constexpr bool STAGE_HOP = ICRL_HALVES_STAGE_IN_HOP || NQ == 4;
  const int jt = w & 3, kh = w >> 2;       // weight gradients / Adam: parameter row block, column half
  const int r = lane & 15, q = lane >> 4;
  const int O = a.L.O, A = a.L.A;
  const int n_out = role == 0 ? A : 1;
  // head outputs along the lane groups (ppo_train_pairs.hip: out_of / pos_of): output o sits at position 4 (o % 4) + o / 4
  auto out_of = [&](int i) { return 4 * i + q; };
  auto pos_of = [](int o) { return 4 * (o & 3) + (o >> 2); };
  const int ro = pos_of(r);
  const int T = a.buf.T, N = a.buf.N;
  const float nu = GPH(a.nu)[0];
  const int n_steps = a.n_steps;
  const PlanStep* __restrict__ const plan_steps = GPH(a.plan_steps);
  const PlanChunk* __restrict__ const plan_chunks = GPH(a.plan_chunks);
  const int* __restrict__ const perms = GPH(a.perms);
  const float* const p_s0 = GPH(role == 0 ? a.buf.log_probs : (role == 1 ? a.buf.reward_values : a.buf.cost_values));
  const float* const p_s1 = GPH(role == 0 ? a.buf.reward_advantages : (role == 1 ? a.buf.reward_returns : a.buf.cost_returns));
  const float* const p_s2 = GPH(a.buf.cost_advantages);
  const float* const p_obs = GPH(a.buf.observations);
  const float* const p_act = GPH(a.buf.actions);
  const int AS = a.buf.act_store;

  // ---- Adam ownership of wave (jt, kh): element (row j = 16 jt + 4 q + i, column k = 16 c + r) of W2 for c in {2 kh, 2 kh + 1}, of W1
  //   for c = kh; kh == 0 only: head-weight element (o = out_of(i), j = 16 jt + r), b1 / b2 entry 16 jt + r (replicated over q, lane
  //   q == 0 stores), wave 0: head bias at position r, wave 1: log_std.  Master weights in LDS, moments and gradients in registers.
  f32x4 mW1, vW1, gW1r, mW2[2], vW2[2], gW2r[2], mWh, vWh, gWhr;
  const int jb = 16 * jt + r;
  float mb1 = 0.f, vb1 = 0.f, mb2 = 0.f, vb2 = 0.f, mex = 0.f, vex = 0.f, gb1r = 0.f, gb2r = 0.f, gex = 0.f;
  int ex_g = -1, ex_s = S::MISC + 63;
  const bool lowk = kh == 0;
  constexpr int W_BH = 0, W_LS = 1;
  auto w1_addr = [&](int i) { return S::W1 + (16 * jt + 4 * q + i) * SX + 16 * kh + r; };
  auto w2_addr = [&](int cc, int i) { return S::W2 + (16 * jt + 4 * q + i) * SH + 16 * (2 * kh + cc) + r; };
  auto store_w1 = [&](const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[w1_addr(i)] = v[i];
  };
  auto load_own_w1 = [&]() -> f32x4 {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sm[w1_addr(i)];
    return v;
  };
  auto store_w2 = [&](int cc, const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[w2_addr(cc, i)] = v[i];
    *reinterpret_cast<f32x4*>(sm + S::W2T + (16 * (2 * kh + cc) + r) * SH + 16 * jt + 4 * q) = v;
  };
  auto load_own_w2 = [&](int cc) -> f32x4 { return lds128(sm + S::W2T + (16 * (2 * kh + cc) + r) * SH + 16 * jt + 4 * q); };
  auto store_wh = [&](const f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[S::WH + (4 * q + i) * SH + 16 * jt + r] = v[i];
    *reinterpret_cast<f32x4*>(sm + S::WHT + (16 * jt + r) * SAH + 4 * q) = v;
  };
  auto load_own_wh = [&]() -> f32x4 { return lds128(sm + S::WHT + (16 * jt + r) * SAH + 4 * q); };
  for (int i = tid; i < S::TOTAL; i += THH) sm[i] = 0.f;
  __syncthreads();
  {
    const PolLayout& L = a.L;
    const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
    const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
    const int gbh = role == 0 ? L.ba : (role == 1 ? L.bv : L.bc);
    {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * jt + 4 * q + i, k = 16 * kh + r;
        const bool mine = k < O;
        pv[i] = mine ? a.params[gW1 + j * O + k] : 0.f;
        mW1[i] = mine ? a.exp_avg[gW1 + j * O + k] : 0.f;
        vW1[i] = mine ? a.exp_avg_sq[gW1 + j * O + k] : 0.f;
      }
      store_w1(pv);
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * jt + 4 * q + i, k = 16 * (2 * kh + cc) + r;
        pv[i] = a.params[gW2 + j * HD + k];
        mW2[cc][i] = a.exp_avg[gW2 + j * HD + k];
        vW2[cc][i] = a.exp_avg_sq[gW2 + j * HD + k];
      }
      store_w2(cc, pv);
    }
    mWh = vWh = gWhr = f32x4{0.f, 0.f, 0.f, 0.f};
    if (lowk) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = out_of(i), j = 16 * jt + r;
        pv[i] = o < n_out ? a.params[gWh + o * HD + j] : 0.f;
        mWh[i] = o < n_out ? a.exp_avg[gWh + o * HD + j] : 0.f;
        vWh[i] = o < n_out ? a.exp_avg_sq[gWh + o * HD + j] : 0.f;
      }
      store_wh(pv);
      if (w == W_BH && ro < n_out) { ex_g = gbh + ro; ex_s = S::BH + r; }
      if (w == W_LS && !DISC && role == 0 && ro < A) { ex_g = L.log_std + ro; ex_s = S::LS + r; }
      if (ex_g >= 0) { mex = a.exp_avg[ex_g]; vex = a.exp_avg_sq[ex_g]; }
      if (q == 0) sm[ex_s] = ex_g >= 0 ? a.params[ex_g] : 0.f;
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
  u64* const xch0 = GPH(a.xch);                                                                    // XCD words of all six workgroups, half 0's norm granules
  u64* const gxp = GPH(a.gx);
  // this part's norm granules (OWNER: ONE area for all four parts — the eight wave granules of a role come from the four owners)
  u64* const nx = (half == 0 || OWNER) ? xch0 : reinterpret_cast<u64*>(reinterpret_cast<char*>(gxp) + ICRL_PPO_SPLIT_BYTES - 512 * half);

  // ---- row stream: the 32 rows of this half are staged by the 512 threads, 16 per row (see ppo_train_pairs.hip for the rules the
  // index / row loads follow: unconditional, clamped, untouched until consumed)
  constexpr bool STAGE_IDLE = ICRL_HALVES_STAGE_IDLE && NQ == 4;
  const int gb_row = STAGE_IDLE ? (tid >> 4) & 15 : tid >> 4, gpart = tid & 15;
  const int gpos = HRQ * half + (gb_row < HRQ ? gb_row : 0);      // position of that row in its 64-row chunk
  const bool stager = STAGE_IDLE ? tid >= 256 : gb_row < HRQ;     // (NQ == 4: 16 rows, staged by the threads of waves 0..3 — STAGE_IDLE: of waves 4..7)
  const bool stager_wave = !STAGE_IDLE || w >= 4;
  constexpr int SW0 = 1;                          // advantage statistics: row stid of the minibatch on waves SW0 .. SW0 + 3
  const int stid = tid - 64 * SW0;
  auto ld_step = [&](int i) -> int4 {
    asm volatile("" : "+v"(i));
    return *reinterpret_cast<const int4*>(plan_steps + i);
  };
  auto ld_chunk = [&](int g) -> int2 {
    asm volatile("" : "+v"(g));
    return *reinterpret_cast<const int2*>(plan_chunks + g);
  };
  auto chunk_idx = [&](const int2& c) -> int { return perms[c.x + (gpos < c.y ? gpos : 0)]; };      // {perm_base, rows}
  auto stat_idx = [&](const int4& p) -> int {
    const int nbp = p.z & NB_MASK;
    return perms[p.w + ((stid >= 0 && stid < nbp) ? stid : 0)];
  };
  constexpr int XRL = OBS > 0 ? (OBS + 15) / 16 : NT1;
  float px[XRL], pact = 0.f, psc = 0.f;
  const float* const p_sc = gpart == 0 ? p_s0 : (gpart == 1 ? p_s1 : p_s2);
  const int sc_dst = gpart == 0 ? S::OLP + gb_row : (gpart == 1 ? S::ADR + gb_row : (gpart == 2 ? S::ADC + gb_row : S::MISC + 62));
  auto issue_rows = [&](int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;
    const unsigned ob = off * (unsigned)O;
#pragma unroll
    for (int i = 0; i < XRL; ++i) { const int k = gpart + 16 * i; px[i] = p_obs[ob + (unsigned)(k < O ? k : O - 1)]; }
    pact = p_act[off * (unsigned)AS + (unsigned)(gpart < AS ? gpart : AS - 1)];      // (the critics fetch the action bytes too: a load is cheaper than a branch here)
    psc = p_sc[off];
  };
  auto commit_rows = [&](int xbase) {
    if (NQ == 4 && !stager) return;
#pragma unroll
    for (int i = 0; i < XRL; ++i) { const int k = gpart + 16 * i; if (OBS > 0 ? k < OBS : k < S::O16) sm[xbase + k * STH + gb_row] = px[i]; }
    if (role == 0) sm[S::ACT + gb_row * SAH + pos_of(gpart)] = gpart < AS ? pact : 0.f;
    sm[sc_dst] = psc;
  };
  float sar = 0.f, sac = 0.f;
  auto issue_stats = [&](int idx) {
    const unsigned off = idx >= 0 ? (unsigned)idx : 0u;
    sar = p_s1[off];
    sac = p_s2[off];
  };
  const bool big_mb = a.hp.batch_size > 128;
  auto stats_partials = [&](int nb) {      // (both halves: the statistics are the whole minibatch's, formed identically)
    if (role != 0 || w < SW0 || w > SW0 + (big_mb ? 3 : 1)) return;
    const bool in = stid < nb;
    const float s_r = wave_sum_fast(in ? sar : 0.f), s_c = wave_sum_fast(in ? sac : 0.f), s_rr = wave_sum_fast(in ? sar * sar : 0.f);
    if (lane == 0) { sm[S::MISC + 3 * (w - SW0)] = s_r; sm[S::MISC + 3 * (w - SW0) + 1] = s_c; sm[S::MISC + 3 * (w - SW0) + 2] = s_rr; }
  };
  float mean_r = 0.f, istd_r = 1.f, mean_c = 0.f;
  auto read_stats = [&](int nb) {
    if (role != 0) return;
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
  auto refresh_gauss = [&]() {   // wave W_LS, lanes q == 0 own log_std r: derived constants of the Gaussian head
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
  // ---- synchronisation inside a quad (the four waves of a row tile: w, w ^ 2, w ^ 4, w ^ 6): a phase counter per wave in LDS; the
  // producer drains its LDS stores and raises its counter, a consumer polls the three others' counters, then reads
  int* const pflag = reinterpret_cast<int*>(sm + S::MISC + 48);      // [8] one word per wave
  int pphase = 0;
  constexpr bool QBAR = ICRL_HALVES_QUAD_BARRIER && NQ == 4 && ICRL_HALVES_LOSS_WAVES >= 4;
  auto quad_signal = [&]() {
    if (QBAR) return;      // (the barrier in quad_wait does both)
    ++pphase;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(pflag + w, pphase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto quad_wait = [&]() {
    if (QBAR) { lds_barrier(); return; }
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
  // lane 0 of wave 7 keeps the running statistics of the role (a kh == 1 wave: no head-weight gradient to form); its sums live in LDS
  const bool book = tid == 64 * 7;
  float* const acc_ent = sm + S::MISC + 56; float* const acc_pg = sm + S::MISC + 57; float* const acc_cf = sm + S::MISC + 58;
  float* const acc_vl = sm + S::MISC + 59; float* const acc_last = sm + S::MISC + 60; float* const acc_kl = sm + S::MISC + 61;
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
  if (tid == 0) sm[S::MISC + 14] = run_on_one_xcd(xch0, slot_j, 3 * NQ, NQ == 4) ? 1.f : 0.f;
  __syncthreads();                      // initial weights visible (refresh_gauss reads log_std)
#ifdef ICRL_ASSUME_XCD_LOCAL      // (measurement only: what a compile-time store scope would buy — no branch per exchange store: 5.88 -> 5.87 us per step: nothing)
  constexpr bool xcd_local = true;
#else
  const bool xcd_local = __builtin_amdgcn_readfirstlane(__float_as_int(sm[S::MISC + 14])) != 0;
#endif
  refresh_gauss();
  int xcur = S::XT0;
  commit_rows(xcur);
  stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  __syncthreads();
  read_stats(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
  const float inv_n_mb = 1.f / (float)((T * N + a.hp.batch_size - 1) / a.hp.batch_size);
  const bool deep_pf = ICRL_HALVES_PREFETCH_AT_ADAM && a.hp.batch_size <= RB;      // one chunk per step: the row stream runs one step deeper
  if (deep_pf) {      // step 1's rows are in flight while step 0 runs (its hop commits them)
    const int idx_now = idx_next;
    idx_next = idx_nx2;
    idx_nx2 = chunk_idx(pc_nx3);
    pc_nx3 = ld_chunk(4);
    issue_rows(idx_now);
  }

  const int b = 16 * rt2 + r;           // this lane's row of the half-chunk (all four q lanes share it)
  float* const pt = sm + (4 * q) * STH + b;     // + image + (16 t + i) STH: element [feature 16 t + 4 q + i][row b]
  constexpr int JT = OBS / 16;
  constexpr bool TAILQ = OBS > 0 && OBS % 16 >= 1 && OBS % 16 <= 4 && JT < NT1;      // ppo_train_pairs.hip: the last K group's components along the lane groups

  // ---- the exchange of the partial gradients: raw 16-byte stores (sc0: the line stays in this XCD's L2 for the partner's L1-bypassing
  // loads; sc1 when the six workgroups do not share an XCD), one flag word per wave behind s_waitcnt vmcnt(0)
  typedef unsigned int raw_u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gxp, 0, (int)ICRL_PPO_SPLIT_BYTES, 0x00020000);
  auto raw_store = [&](int byte_off, const f32x4& v) {
    const raw_u4 u = __builtin_bit_cast(raw_u4, v);
    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(u, grs, byte_off, 0, 1);
    else __builtin_amdgcn_raw_buffer_store_b128(u, grs, byte_off, 0, 16);
  };
  auto raw_load = [&](int byte_off) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, byte_off, 0, 16)); };
  // the same with the address split into the lane's slot (one loop-invariant register) + a UNIFORM offset in the instruction's scalar-offset operand: no
  // vector add per access (late round 6; ICRL_HALVES_SOFFSET)
  const int lane_slot = tid * 16;
  auto raw_store_u = [&](int uni_off, const f32x4& v) {
    if (!ICRL_HALVES_SOFFSET) { raw_store(uni_off + lane_slot, v); return; }
    const raw_u4 u = __builtin_bit_cast(raw_u4, v);
    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(u, grs, lane_slot, uni_off, 1);
    else __builtin_amdgcn_raw_buffer_store_b128(u, grs, lane_slot, uni_off, 16);
  };
  auto raw_load_u = [&](int uni_off) -> f32x4 {
    if (!ICRL_HALVES_SOFFSET) return raw_load(uni_off + lane_slot);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, lane_slot, uni_off, 16));
  };

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
    ps_nx3 = ld_step(st + 3 < n_steps + 2 ? st + 3 : n_steps + 1);
    const int nb = ps.nb_flags & NB_MASK;
    const float inv_nb = __builtin_amdgcn_rcpf((float)nb);
    const float c_mean_r = mean_r, c_mean_c = mean_c, c_istd_r = istd_r;
    const float cpol_nb = inv_nb * __builtin_amdgcn_rcpf(1.f + nu);
    issue_stats(sidx_next);
    sidx_next = stat_idx(ps_nx2);
    float mb_s0 = 0.f, mb_s1 = 0.f, mb_s2 = 0.f, mb_s3 = 0.f, mb_s4 = 0.f;
    const int xrole = ((int)(step & 1) * 3 + role) * NQ * HX_BLK;      // the NQ blocks of this role and step parity
    const int xmine = xrole + half * HX_BLK, xtheirs = xrole + (1 - half) * HX_BLK;      // (xtheirs: NQ == 2)

    const int n_chunks = (nb + RB - 1) / RB;
    for (int ch = 0; ch < n_chunks; ++ch, ++g_chunk) {
      const int nrows = (nb - ch * RB) < RB ? (nb - ch * RB) : RB;
      if (ch > 0) {
        xcur = xcur == S::XT0 ? S::XT1 : S::XT0;
        commit_rows(xcur);
        lds_barrier();                    // a row is staged by threads of several waves
      }
      const bool valid = HRQ * half + b < nrows;
      // ================= forward =================
      f32x4 h1c = f32x4{0.f, 0.f, 0.f, 0.f}, h2c = h1c, outc = h1c;               // own feature tile t = fq
      if (fwd_wave) {  // layer 1
        float bx[NT1][4];                 // x[row b][k = 16 js + 4 q + e]
        const float* pb = sm + xcur + (4 * q) * STH + b;
#pragma unroll
        for (int js = 0; js < NT1; ++js)
#pragma unroll
          for (int e = 0; e < 4; ++e) bx[js][e] = ((OBS == 0 || 16 * js + e < OBS) && !(TAILQ && js == JT)) ? pb[(16 * js + e) * STH] : 0.f;
        const float bt = TAILQ ? sm[xcur + (16 * JT + q) * STH + b] : 0.f;      // x[row b][k = 16 JT + q]
        const float* pa = sm + S::W1 + (16 * fq + r) * SX + 4 * q;
        f32x4 aw[NT1];
#pragma unroll
        for (int js = 0; js < NT1; ++js)
          if (!(TAILQ && js >= JT)) aw[js] = lds128(pa + 16 * js);
        const float at = TAILQ ? sm[S::W1 + (16 * fq + r) * SX + 16 * JT + q] : 0.f;
        f32x4 z = lds128(sm + S::B1 + 16 * fq + 4 * q);      // the bias is the accumulator's initial value
        HFENCE();
#pragma unroll
        for (int js = 0; js < NT1; ++js)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((OBS == 0 || 16 * js + e < OBS) && !(TAILQ && js >= JT)) z = MFMA_F32(aw[js][e], bx[js][e], z);
        if (TAILQ) z = MFMA_F32(at, bt, z);
#pragma unroll
        for (int i = 0; i < 4; ++i) h1c[i] = fast_tanh(z[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::H1T + (16 * fq + i) * STH] = h1c[i];
        *reinterpret_cast<f32x4*>(sm + S::H1R + b * SRM + 16 * fq + 4 * q) = h1c;
      }
      auto prefetch_next = [&]() {  // the next chunk's rows (random pieces of the rollout buffer: several microseconds away)
        const int idx_now = idx_next;
        idx_next = idx_nx2;
        idx_nx2 = chunk_idx(pc_nx3);
        pc_nx3 = ld_chunk(g_chunk + 4);
        issue_rows(idx_now);
      };
      const bool pf_here = !deep_pf;      // (deep_pf: at the top of Adam)
      if (pf_here && stager_wave && (!ICRL_HALVES_LATE_PREFETCH || !fwd_wave)) prefetch_next();
      f32x4 dout = f32x4{0.f, 0.f, 0.f, 0.f};
      float pl_olp = 0.f, pl_adr = 0.f, pl_adc = 0.f;      // (ICRL_HALVES_LOSS_PRELOAD)
      f32x4 pl_act = dout, pl_iv = dout, pl_hiv = dout, pl_lsd = dout, pl_wht = dout;
      if (fwd_wave) {      // ---- the rest of forward, the loss tail and the activation backward: the waves of the row tiles (NQ == 4: waves 0..3)
      quad_signal();               // (P1) this wave's features of h1 are complete
      {  // layer 2: own quarter of K from registers, the other three from the row-major image
        const float* pa = sm + S::W2 + (16 * fq + r) * SH + 4 * q;
        const f32x4 awo = lds128(pa + 16 * fq);
        f32x4 awp[3];
#pragma unroll
        for (int d = 1; d < 4; ++d) awp[d - 1] = lds128(pa + 16 * ((fq + d) & 3));
        f32x4 z = lds128(sm + S::B2 + 16 * fq + 4 * q);
        if (ICRL_HALVES_LATE_PREFETCH && pf_here) { HFENCE(); prefetch_next(); HFENCE(); }
#pragma unroll
        for (int e = 0; e < 4; ++e) z = MFMA_F32(awo[e], h1c[e], z);
        quad_wait();               // the other three waves' features of h1 are complete
        f32x4 hp[3];
        const float* ph1 = sm + S::H1R + b * SRM + 4 * q;
#pragma unroll
        for (int d = 1; d < 4; ++d) hp[d - 1] = lds128(ph1 + 16 * ((fq + d) & 3));
        HFENCE();
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) z = MFMA_F32(awp[d][e], hp[d][e], z);
#pragma unroll
        for (int i = 0; i < 4; ++i) h2c[i] = fast_tanh(z[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::H2T + (16 * fq + i) * STH] = h2c[i];
      }
      {  // head, split over K four ways; partial tiles exchanged through LDS
        const f32x4 aw = lds128(sm + S::WH + r * SH + 16 * fq + 4 * q);
        f32x4 acc = fq == 0 ? lds128(sm + S::BH + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = MFMA_F32(aw[e], h2c[e], acc);
        float* const hpx = sm + S::HPX + (rt2 * 4 * 64 + lane) * 4;
        *reinterpret_cast<f32x4*>(hpx + fq * 256) = acc;
        if (ICRL_HALVES_LOSS_PRELOAD) {
          pl_olp = sm[S::OLP + b]; pl_adr = sm[S::ADR + b]; pl_adc = sm[S::ADC + b];
          if (role == 0 && !DISC) {
            pl_act = lds128(sm + S::ACT + b * SAH + 4 * q);
            pl_iv = lds128(sm + S::GAU + 4 * q); pl_hiv = lds128(sm + S::GAU + 16 + 4 * q); pl_lsd = lds128(sm + S::GAU + 32 + 4 * q);
          }
          pl_wht = lds128(sm + S::WHT + (16 * fq + r) * SAH + 4 * q);
        }
        quad_signal(); quad_wait();  // (P3) all four partial tiles stored
        const f32x4 p0 = lds128(hpx), p1 = lds128(hpx + 256), p2 = lds128(hpx + 512), p3 = lds128(hpx + 768);
#pragma unroll
        for (int i = 0; i < 4; ++i) outc[i] = (p0[i] + p1[i]) + (p2[i] + p3[i]);
      }
      STAMP(0)   // forward
      // ============ loss + d loss / d head output (all four waves of a quad: identical values) ============
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
      f32x4 dls_p = f32x4{0.f, 0.f, 0.f, 0.f};      // (ICRL_HALVES_DEFER_STATS: dlp * d log-prob / d log_std per element, reduced behind P4)
      constexpr bool DEFER = ICRL_HALVES_DEFER_STATS && ICRL_HALVES_LOSS_WAVES >= 4;
      if (ICRL_HALVES_LOSS_WAVES >= 4 || fq < ICRL_HALVES_LOSS_WAVES) {
        if (role == 0) {
          const int ngp = (A + 3) >> 2;      // groups of four outputs that hold real ones
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
            const int act = (int)sm[S::ACT + b * SAH];
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
            const f32x4 actv = ICRL_HALVES_LOSS_PRELOAD ? pl_act : lds128(sm + S::ACT + b * SAH + 4 * q);
            const f32x4 iv = ICRL_HALVES_LOSS_PRELOAD ? pl_iv : lds128(sm + S::GAU + 4 * q), hiv = ICRL_HALVES_LOSS_PRELOAD ? pl_hiv : lds128(sm + S::GAU + 16 + 4 * q),
                        lsd = ICRL_HALVES_LOSS_PRELOAD ? pl_lsd : lds128(sm + S::GAU + 32 + 4 * q);
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
          const float old_lp = ICRL_HALVES_LOSS_PRELOAD ? pl_olp : sm[S::OLP + b];
          const float ratio = __expf(lp - old_lp);
          const float Ar = ((ICRL_HALVES_LOSS_PRELOAD ? pl_adr : sm[S::ADR + b]) - c_mean_r) * c_istd_r;
          const float Ac = (ICRL_HALVES_LOSS_PRELOAD ? pl_adc : sm[S::ADC + b]) - c_mean_c;
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
            auto dls = [&](int i) { dout[i] = dlp * g1[i]; if (DEFER) dls_p[i] = dlp * g2[i]; else t[i] = row_sum(dlp * g2[i]); };
            dls(0);
            if (ngp > 1) { dls(1); if (ngp > 2) { dls(2); dls(3); } }
            if (!DEFER && fq == 0 && r == 0) *reinterpret_cast<f32x4*>(sm + S::PLS + 16 * rt2 + 4 * q) = t;
          }
          const bool cnt = valid && q == 0;
          v0 = cnt ? fminf(s1, s2) : 0.f; v1 = cnt ? Ac * ratio : 0.f; v2 = (cnt && fabsf(ratio - 1.f) > clip) ? 1.f : 0.f;
          v3 = cnt ? old_lp - lp : 0.f; v4 = cnt ? ent : 0.f;
        } else {
          const float v = quad_rows_sum(q == 0 ? outc[0] : 0.f);
          const float R = ICRL_HALVES_LOSS_PRELOAD ? pl_adr : sm[S::ADR + b];
          float vp = v, pass = 1.f;
          if (vclip >= 0.f) {
            const float old = ICRL_HALVES_LOSS_PRELOAD ? pl_olp : sm[S::OLP + b];
            const float dv = v - old;
            vp = old + fminf(fmaxf(dv, -vclip), vclip);
            pass = (dv >= -vclip && dv <= vclip) ? 1.f : 0.f;
          }
          const float e = vp - R;
          const float d0 = valid ? vcoef * 2.f * e * inv_nb * pass : 0.f;
          dout[0] = q == 0 ? d0 : 0.f;
          v0 = (valid && q == 0) ? e * e : 0.f;
        }
        if (!DEFER && fq == 0) {     // one wave of the quad reports the tile's statistics
          v0 = row_sum(v0); v1 = row_sum(v1); v2 = row_sum(v2); v3 = row_sum(v3);
          if (DISC) v4 = row_sum(v4);
          if (lane == 0) { float* pst = sm + S::PST + 8 * rt2; pst[0] = v0; pst[1] = v1; pst[2] = v2; pst[3] = v3; pst[4] = v4; }
        }
      }
#if ICRL_HALVES_LOSS_WAVES < 4
      {  // (P3b) the quad's first wave hands d loss / d head output over, lane for lane; the waves that skipped the tail wait for it
        float* const dox = sm + S::DOX + (rt2 * 64 + lane) * 4;
        if (fq == 0) *reinterpret_cast<f32x4*>(dox) = dout;
        quad_signal();
        if (fq >= ICRL_HALVES_LOSS_WAVES) { quad_wait(); dout = lds128(dox); }
      }
#endif
      STAMP(1)   // loss
      // ================= backward of the activations =================
      f32x4 dz2c, dz1c;
      {  // dH2^T = Wh^T . dOut^T for the own feature tile: A = WHT[j = 16 fq + r][position 4 q + e]; MFMA e covers the outputs 4 e .. 4 e + 3
        const int ng = (n_out + 3) >> 2;
        const f32x4 aw = ICRL_HALVES_LOSS_PRELOAD ? pl_wht : lds128(sm + S::WHT + (16 * fq + r) * SAH + 4 * q);
        f32x4 acc = MFMA_F32(aw[0], dout[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
        if (ng > 1) {
          acc = MFMA_F32(aw[1], dout[1], acc);
          if (ng > 2) { acc = MFMA_F32(aw[2], dout[2], acc); acc = MFMA_F32(aw[3], dout[3], acc); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dz2c[i] = fmaf(-(h2c[i] * h2c[i]), acc[i], acc[i]);   // acc (1 - h2^2)
#pragma unroll
        for (int i = 0; i < 4; ++i) pt[S::DZ2T + (16 * fq + i) * STH] = dz2c[i];
        *reinterpret_cast<f32x4*>(sm + S::DZ2R + b * SRM + 16 * fq + 4 * q) = dz2c;
        if (fq == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pt[S::DOT + i * STH] = dout[i];
        }
      }
      quad_signal();               // (P4) this wave's features of dz2 complete
      if (DEFER && fq == 0) {      // the tile's logged statistics and d log_std partial (read behind the barrier (S5))
        if (role == 0 && !DISC) {
          f32x4 t;
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = row_sum(dls_p[i]);
          if (r == 0) *reinterpret_cast<f32x4*>(sm + S::PLS + 16 * rt2 + 4 * q) = t;
        }
        v0 = row_sum(v0); v1 = row_sum(v1); v2 = row_sum(v2); v3 = row_sum(v3);
        if (DISC) v4 = row_sum(v4);
        if (lane == 0) { float* pst = sm + S::PST + 8 * rt2; pst[0] = v0; pst[1] = v1; pst[2] = v2; pst[3] = v3; pst[4] = v4; }
      }
      {  // dH1^T = W2^T . dz2^T: A = W2T[k = 16 fq + r][j = 16 js + 4 q + e]; own quarter of K before the wait for the others
        const float* pa = sm + S::W2T + (16 * fq + r) * SH + 4 * q;
        const f32x4 awo = lds128(pa + 16 * fq);
        f32x4 awp[3];
#pragma unroll
        for (int d = 1; d < 4; ++d) awp[d - 1] = lds128(pa + 16 * ((fq + d) & 3));
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = MFMA_F32(awo[e], dz2c[e], acc);
        quad_wait();               // the other three waves' features of dz2 complete
        f32x4 dp[3];
        const float* pz = sm + S::DZ2R + b * SRM + 4 * q;
#pragma unroll
        for (int d = 1; d < 4; ++d) dp[d - 1] = lds128(pz + 16 * ((fq + d) & 3));
        HFENCE();
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = MFMA_F32(awp[d][e], dp[d][e], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) dz1c[i] = fmaf(-(h1c[i] * h1c[i]), acc[i], acc[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) pt[S::DZ1T + (16 * fq + i) * STH] = dz1c[i];
      } else if (QBAR) {      // waves 4..7: the quad's three hand-off barriers
        lds_barrier(); lds_barrier(); lds_barrier();
      }      // fwd_wave
      STAMP(2)   // activation backward
      if (ch == 0) {   // gradient accumulators start their life here
        gW1r = gW2r[0] = gW2r[1] = gWhr = f32x4{0.f, 0.f, 0.f, 0.f};
        gb1r = 0.f; gb2r = 0.f; gex = 0.f;
      }
      lds_barrier();  // (S5) every quad's columns of h1^T, h2^T, dz1^T, dz2^T, dOut^T (and the loss partials) are complete
      // ================= weight gradients (K = this part's HRQ rows) =================
      const bool last_chunk = ch + 1 == n_chunks;
      {  // dW2 rows 16 jt.., column tiles 2 kh, 2 kh + 1
        f32x4 az[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // dz2^T[j = 16 jt + r][rows 16 js + 4 q + e]
        const float* pa = sm + S::DZ2T + (16 * jt + r) * STH + 4 * q;
#pragma unroll
        for (int js = 0; js < NJS; ++js) az[js] = lds128(pa + 16 * js);
        f32x4 bh[2][2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const float* pb = sm + S::H1T + (16 * (2 * kh + cc) + r) * STH + 4 * q;
#pragma unroll
          for (int js = 0; js < NJS; ++js) bh[cc][js] = lds128(pb + 16 * js);
        }
        HFENCE();
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
          for (int js = 0; js < NJS; ++js)
#pragma unroll
            for (int e = 0; e < 4; ++e) gW2r[cc] = MFMA_F32(az[js][e], bh[cc][js][e], gW2r[cc]);
        }
        if (lowk) {
          const float s = ((az[0][0] + az[0][1]) + (az[0][2] + az[0][3])) + ((az[1][0] + az[1][1]) + (az[1][2] + az[1][3]));     // d b2[16 jt + r]
          gb2r += quad_rows_sum(s);
        }
      }
      {  // dW1 rows 16 jt.., observation tile kh
        f32x4 az[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // dz1^T[j = 16 jt + r][rows]
        const float* pa = sm + S::DZ1T + (16 * jt + r) * STH + 4 * q;
#pragma unroll
        for (int js = 0; js < NJS; ++js) az[js] = lds128(pa + 16 * js);
        const float* pb = sm + xcur + (16 * kh + r) * STH + 4 * q;     // x^T[k][rows 16 js + 4 q + e]
        f32x4 bx[2];
#pragma unroll
        for (int js = 0; js < NJS; ++js) bx[js] = lds128(pb + 16 * js);
        HFENCE();
#pragma unroll
        for (int js = 0; js < NJS; ++js)
#pragma unroll
          for (int e = 0; e < 4; ++e) gW1r = MFMA_F32(az[js][e], bx[js][e], gW1r);
        // observation pad columns (k >= obs): X holds unmasked fill there; their weights, gradients and moments stay 0
        if (last_chunk && 16 * kh + r >= O) gW1r = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lowk) {
          const float s = ((az[0][0] + az[0][1]) + (az[0][2] + az[0][3])) + ((az[1][0] + az[1][1]) + (az[1][2] + az[1][3]));
          gb1r += quad_rows_sum(s);
        }
      }
      // the W2 tiles go out to the other half while the remaining GEMMs run (their MFMA chains are complete by now)
      if (ICRL_HALVES_EARLY_PUBLISH && last_chunk) {
        raw_store_u(xmine + 1 * THH * 16, gW2r[0]);
        raw_store_u(xmine + 2 * THH * 16, gW2r[1]);
      }
      if (lowk) {   // dWh columns 16 jt..: A = dOut^T[position r][rows], B = h2^T[j = 16 jt + r][rows]
        f32x4 ao[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, bh[2];
        const float* pa = sm + S::DOT + r * STH + 4 * q;
        const float* pb = sm + S::H2T + (16 * jt + r) * STH + 4 * q;
#pragma unroll
        for (int js = 0; js < NJS; ++js) { ao[js] = lds128(pa + 16 * js); bh[js] = lds128(pb + 16 * js); }
        HFENCE();
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int js = 0; js < NJS; ++js)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[js] = MFMA_F32(ao[js][e], bh[js][e], acc[js]);
#pragma unroll
        for (int i = 0; i < 4; ++i) gWhr[i] += acc[0][i] + acc[1][i];
        float s = ((ao[0][0] + ao[0][1]) + (ao[0][2] + ao[0][3])) + ((ao[1][0] + ao[1][1]) + (ao[1][2] + ao[1][3]));      // head bias: row sums of dOut^T
        s = quad_rows_sum(s);
        const float sl = sm[S::PLS + r] + sm[S::PLS + 16 + r];       // d log_std at position r = the two row tiles' partials
        gex += w == W_BH ? s : ((!DISC && role == 0 && w == W_LS) ? sl : 0.f);
      }
      if (book) {
        mb_s0 += sm[S::PST + 0] + sm[S::PST + 8];
        mb_s1 += sm[S::PST + 1] + sm[S::PST + 9];
        mb_s2 += sm[S::PST + 2] + sm[S::PST + 10];
        mb_s3 += sm[S::PST + 3] + sm[S::PST + 11];
        if (DISC) mb_s4 += sm[S::PST + 4] + sm[S::PST + 12];
      }
      if (!last_chunk) lds_barrier();  // chunk buffers free
      STAMP(3)   // weight gradients
    }  // chunks

    // ================= partial gradients of this half <-> the other half of the same network =================
#if ICRL_HALVES_NORM_LOOK_EARLY
    u64 v_early = 0;
#endif
    {
      f32x4 gsc = f32x4{gb1r, gb2r, gex, 0.f};
      if (!ICRL_HALVES_EARLY_PUBLISH) {
        raw_store_u(xmine + 1 * THH * 16, gW2r[0]);
        raw_store_u(xmine + 2 * THH * 16, gW2r[1]);
      }
      raw_store_u(xmine + 0 * THH * 16, gW1r);
      if (lowk) {
        raw_store_u(xmine + 3 * THH * 16, gWhr);
        raw_store_u(xmine + 4 * THH * 16, gsc);
      }
      if (book) {
        raw_store_u(xmine + 5 * THH * 16, f32x4{mb_s0, mb_s1, mb_s2, mb_s3});
        if (DISC) raw_store_u(xmine + 6 * THH * 16, f32x4{mb_s4, 0.f, 0.f, 0.f});
      }
      // every store of this wave has been acknowledged (it is in the L2 the partner reads through, or beyond) -> the wave's flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        if (xcd_local) __builtin_amdgcn_raw_buffer_store_b32(step, grs, xmine + HX_FLAG + 64 * w, 0, 1);
        else __builtin_amdgcn_raw_buffer_store_b32(step, grs, xmine + HX_FLAG + 64 * w, 0, 16);
      }
#if ICRL_HALVES_ADAM_PRE
      {  // beta-scaled moments: needs neither the summed gradient nor the clip coefficient (completed by this step's Adam below)
        const float omw1_ = 1.f - w1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { mW1[i] *= omw1_; vW1[i] *= adam_b2f; mW2[0][i] *= omw1_; vW2[0][i] *= adam_b2f; mW2[1][i] *= omw1_; vW2[1][i] *= adam_b2f; }
        if (lowk) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { mWh[i] *= omw1_; vWh[i] *= adam_b2f; }
          mb1 *= omw1_; vb1 *= adam_b2f; mb2 *= omw1_; vb2 *= adam_b2f; mex *= omw1_; vex *= adam_b2f;
        }
      }
#endif
#if ICRL_HALVES_FIRST_LOOK
      // (four parts) the first look at the three peers' flags is ISSUED before the staging and read behind it
      unsigned fl_a = 0, fl_b = 0, fl_c = 0;
      if (NQ == 4 && owner) {
        const int fo = xrole + HX_FLAG + 64 * w;
        fl_a = __builtin_amdgcn_raw_buffer_load_b32(grs, fo + ((half + 1) & 3) * HX_BLK, 0, 16);
        fl_b = __builtin_amdgcn_raw_buffer_load_b32(grs, fo + ((half + 2) & 3) * HX_BLK, 0, 16);
        fl_c = __builtin_amdgcn_raw_buffer_load_b32(grs, fo + ((half + 3) & 3) * HX_BLK, 0, 16);
      }
#endif
      if (STAGE_HOP) {
        commit_rows(xcur == S::XT0 ? S::XT1 : S::XT0);
        stats_partials(__builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK);
      }
      bool timed_out = false;
#if ICRL_HALVES_NORM_LOOK_EARLY
      // (four parts) the first look at the OTHER networks' norm granules is issued here, in front of the partial-gradient sums: the critics' workgroups
      // are 1-2 k cycles ahead of the policy's, whose own look then costs no trip of its own
      if (NQ == 4 && tid < 24 && (tid >> 3) != role) v_early = __hip_atomic_load(nx + (step & 1) * 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      if constexpr (NQ == 4) {
        // four parts: every workgroup forms (q0 + q1) + (q2 + q3) from the three others' blocks and its own registers — the same floats in the same
        // order on all four, so they stay replicas.  One pair at a time (register pressure): flags of this wave's peers, then their groups.
#if ICRL_HALVES_FLAGS_TOGETHER
        // the flags of this wave's THREE peers in flight together: three blocking looks in a row cost three trips through the L2 even when all
        // three flags are long set; the groups are still fetched pair by pair (register pressure)
        if (owner) {
          const int fo = xrole + HX_FLAG + 64 * w;
          const int ka = (half + 1) & 3, kb = (half + 2) & 3, kc = (half + 3) & 3;
          int spins = 0;
#if ICRL_HALVES_FIRST_LOOK
          const bool first_ok = fl_a == step && fl_b == step && fl_c == step;
#else
          const bool first_ok = false;
#endif
          while (!first_ok) {
            const unsigned fa = __builtin_amdgcn_raw_buffer_load_b32(grs, fo + ka * HX_BLK, 0, 16);
            const unsigned fb = __builtin_amdgcn_raw_buffer_load_b32(grs, fo + kb * HX_BLK, 0, 16);
            const unsigned fc = __builtin_amdgcn_raw_buffer_load_b32(grs, fo + kc * HX_BLK, 0, 16);
            if (((fa ^ step) | (fb ^ step) | (fc ^ step)) == 0u) break;      // (no short-circuit: `fa == step && ...` lets the compiler sink the second and third look behind the first compare)
            if (++spins >= (1 << 22)) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          asm volatile("" ::: "memory");
        }
        auto wait_flag = [&](int) {};
#else
        auto wait_flag = [&](int k) {
          int spins = 0;
          while (true) {
            const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(grs, xrole + k * HX_BLK + HX_FLAG + 64 * w, 0, 16);
            if (f == step) break;
            if (++spins >= (1 << 22)) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          asm volatile("" ::: "memory");
        };
#endif
#if ICRL_HALVES_PARTNER_SUM && !ICRL_HALVES_OWNER_ADAM && ICRL_HALVES_FLAGS_TOGETHER
        {
          (void)wait_flag;
          const int xa = xrole + (half ^ 1) * HX_BLK, xb = xrole + (half ^ 2) * HX_BLK, xc = xrole + (half ^ 3) * HX_BLK;
          const f32x4 a0 = raw_load_u(xa + 0 * THH * 16), b0 = raw_load_u(xb + 0 * THH * 16), c0 = raw_load_u(xc + 0 * THH * 16);
          const f32x4 a1 = raw_load_u(xa + 1 * THH * 16), b1 = raw_load_u(xb + 1 * THH * 16), c1 = raw_load_u(xc + 1 * THH * 16);
          const f32x4 a2 = raw_load_u(xa + 2 * THH * 16), b2 = raw_load_u(xb + 2 * THH * 16), c2 = raw_load_u(xc + 2 * THH * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            gW1r[i] = (gW1r[i] + a0[i]) + (b0[i] + c0[i]);
            gW2r[0][i] = (gW2r[0][i] + a1[i]) + (b1[i] + c1[i]);
            gW2r[1][i] = (gW2r[1][i] + a2[i]) + (b2[i] + c2[i]);
          }
          if (lowk) {      // (wave-uniform)
            const f32x4 a3 = raw_load_u(xa + 3 * THH * 16), b3 = raw_load_u(xb + 3 * THH * 16), c3 = raw_load_u(xc + 3 * THH * 16);
            const f32x4 a4 = raw_load_u(xa + 4 * THH * 16), b4 = raw_load_u(xb + 4 * THH * 16), c4 = raw_load_u(xc + 4 * THH * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              gWhr[i] = (gWhr[i] + a3[i]) + (b3[i] + c3[i]);
              gsc[i] = (gsc[i] + a4[i]) + (b4[i] + c4[i]);
            }
            gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
          }
          if (w == 7) {    // the book-keeping lane's wave: every lane fetches (lane 0's slots hold the sums; the other lanes' words are never written and stay zero)
            const f32x4 a5 = raw_load_u(xa + 5 * THH * 16), b5 = raw_load_u(xb + 5 * THH * 16), c5 = raw_load_u(xc + 5 * THH * 16);
            mb_s0 = (mb_s0 + a5[0]) + (b5[0] + c5[0]); mb_s1 = (mb_s1 + a5[1]) + (b5[1] + c5[1]);
            mb_s2 = (mb_s2 + a5[2]) + (b5[2] + c5[2]); mb_s3 = (mb_s3 + a5[3]) + (b5[3] + c5[3]);
            if (DISC) {
              const f32x4 a6 = raw_load_u(xa + 6 * THH * 16), b6 = raw_load_u(xb + 6 * THH * 16), c6 = raw_load_u(xc + 6 * THH * 16);
              mb_s4 = (mb_s4 + a6[0]) + (b6[0] + c6[0]);
            }
          }
        }
#else
        struct Part { f32x4 g0, g1, g2, g3, g4, b0, b1; };
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        const Part own = {gW1r, gW2r[0], gW2r[1], gWhr, gsc, f32x4{mb_s0, mb_s1, mb_s2, mb_s3}, f32x4{mb_s4, 0.f, 0.f, 0.f}};
        auto fetch = [&](int k) -> Part {
          if (k == half) return own;
          wait_flag(k);
          const int xb = xrole + k * HX_BLK;
          Part p = {raw_load(xb + (0 * THH + tid) * 16), raw_load(xb + (1 * THH + tid) * 16), raw_load(xb + (2 * THH + tid) * 16), z4, z4, z4, z4};
          if (lowk) { p.g3 = raw_load(xb + (3 * THH + tid) * 16); p.g4 = raw_load(xb + (4 * THH + tid) * 16); }
          if (book) { p.b0 = raw_load(xb + (5 * THH + tid) * 16); if (DISC) p.b1 = raw_load(xb + (6 * THH + tid) * 16); }
          return p;
        };
        auto add = [&](const Part& x, const Part& y) -> Part {
          Part r;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            r.g0[i] = x.g0[i] + y.g0[i]; r.g1[i] = x.g1[i] + y.g1[i]; r.g2[i] = x.g2[i] + y.g2[i]; r.g3[i] = x.g3[i] + y.g3[i];
            r.g4[i] = x.g4[i] + y.g4[i]; r.b0[i] = x.b0[i] + y.b0[i]; r.b1[i] = x.b1[i] + y.b1[i];
          }
          return r;
        };
        if (owner) {      // (OWNER: the other three parts keep their partials — they receive this wave's updated parameters instead)
          const Part s01 = add(fetch(0), fetch(1));
          const Part s23 = add(fetch(2), fetch(3));
          const Part t = add(s01, s23);
          gW1r = t.g0; gW2r[0] = t.g1; gW2r[1] = t.g2; gWhr = t.g3; gsc = t.g4;
          gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
          mb_s0 = t.b0[0]; mb_s1 = t.b0[1]; mb_s2 = t.b0[2]; mb_s3 = t.b0[3]; mb_s4 = t.b1[0];
        }
#endif
      } else {
#if ICRL_HALVES_POLL_ROLL
      unsigned pq0, pq1, pq2, pq3;
#endif
      {
        int spins = 0;
#if ICRL_HALVES_POLL_ROLL
        // FOUR looks at the partner's flag in flight, ~130 cycles apart, each re-issued as it returns: the flag is seen within a quarter of a
        // trip through the L2 of its arrival instead of within a whole one (a look that just missed costs the next look's full round trip).
        // Hand-written: the compiler's waitcnt pass waits for ALL outstanding loads at a loop header.  Loads return in order, so
        // `vmcnt(3)` = the oldest look is back whatever else is in flight.  Up to three looks are still outstanding at the exit; they are
        // older than the data loads below, so every wait the compiler inserts for those covers them — their registers are kept allocated
        // until then by the empty asm behind the sums.
        typedef int rsrc4 __attribute__((ext_vector_type(4)));
        const unsigned long long gbase = (unsigned long long)gxp;
        rsrc4 rs4;
        rs4[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)gbase);
        rs4[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((gbase >> 32) & 0xffffu));
        rs4[2] = (int)ICRL_PPO_SPLIT_BYTES;
        rs4[3] = 0x00020000;
        const int foff = xtheirs + HX_FLAG + 64 * w;
        int left = 1 << 20;
        asm volatile(
            "buffer_load_dword %0, %5, %6, 0 offen sc1\n\t"
            "s_sleep 2\n\t"
            "buffer_load_dword %1, %5, %6, 0 offen sc1\n\t"
            "s_sleep 2\n\t"
            "buffer_load_dword %2, %5, %6, 0 offen sc1\n\t"
            "s_sleep 2\n\t"
            "buffer_load_dword %3, %5, %6, 0 offen sc1\n\t"
            "1:\n\t"
            "s_waitcnt vmcnt(3)\n\t"
            "v_cmp_eq_u32_e32 vcc, %7, %0\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "buffer_load_dword %0, %5, %6, 0 offen sc1\n\t"
            "s_waitcnt vmcnt(3)\n\t"
            "v_cmp_eq_u32_e32 vcc, %7, %1\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "buffer_load_dword %1, %5, %6, 0 offen sc1\n\t"
            "s_waitcnt vmcnt(3)\n\t"
            "v_cmp_eq_u32_e32 vcc, %7, %2\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "buffer_load_dword %2, %5, %6, 0 offen sc1\n\t"
            "s_waitcnt vmcnt(3)\n\t"
            "v_cmp_eq_u32_e32 vcc, %7, %3\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "buffer_load_dword %3, %5, %6, 0 offen sc1\n\t"
            "s_sub_i32 %4, %4, 1\n\t"
            "s_cmp_gt_i32 %4, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            : "=&v"(pq0), "=&v"(pq1), "=&v"(pq2), "=&v"(pq3), "+s"(left)
            : "v"(foff), "s"(rs4), "s"(step)
            : "vcc", "scc", "memory");
        timed_out = left <= 0;
        (void)spins;
#else
        while (true) {
          const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(grs, xtheirs + HX_FLAG + 64 * w, 0, 16);
          if (f == step) break;
          if (++spins >= (1 << 22)) { timed_out = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#endif
      }
      asm volatile("" ::: "memory");
      const f32x4 c0 = raw_load(xtheirs + (0 * THH + tid) * 16), c1 = raw_load(xtheirs + (1 * THH + tid) * 16), c2 = raw_load(xtheirs + (2 * THH + tid) * 16);
      f32x4 c3 = f32x4{0.f, 0.f, 0.f, 0.f}, c4 = c3, c5 = c3, c6 = c3;
      if (lowk) { c3 = raw_load(xtheirs + (3 * THH + tid) * 16); c4 = raw_load(xtheirs + (4 * THH + tid) * 16); }
      if (book) { c5 = raw_load(xtheirs + (5 * THH + tid) * 16); if (DISC) c6 = raw_load(xtheirs + (6 * THH + tid) * 16); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { gW1r[i] += c0[i]; gW2r[0][i] += c1[i]; gW2r[1][i] += c2[i]; gWhr[i] += c3[i]; gsc[i] += c4[i]; }     // own + partner (commutative: both halves agree)
#if ICRL_HALVES_POLL_ROLL
      asm volatile("" :: "v"(pq0), "v"(pq1), "v"(pq2), "v"(pq3), "v"(gW2r[1][3]));      // (the looks' registers stay allocated until the data is in)
#endif
      gb1r = gsc[0]; gb2r = gsc[1]; gex = gsc[2];
      mb_s0 += c5[0]; mb_s1 += c5[1]; mb_s2 += c5[2]; mb_s3 += c5[3]; mb_s4 += c6[0];
      }      // NQ == 2
      if (timed_out) sm[S::MISC + 13] = 1.f;       // reported through the status word like a timed-out norm exchange
    }

    // entropy term of the Gaussian policy loss: d(ent_coef * -mean(H)) / d log_std = -ent_coef (once, on the summed gradient)
    if (!DISC && role == 0 && w == W_LS && ro < A) gex += -ent_coef;

    // ================= global gradient norm: this wave's partial sum of squares -> its own 8-byte granule =================
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ss = fmaf(gW1r[i], gW1r[i], ss);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int i = 0; i < 4; ++i) ss = fmaf(gW2r[cc][i], gW2r[cc][i], ss);
    {
      float sb = 0.f;      // per-row entries (replicated over the lane groups: counted on q == 0)
      if (lowk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ss = fmaf(gWhr[i], gWhr[i], ss);
        sb = ex_g >= 0 ? gex * gex : 0.f;
        sb = fmaf(gb1r, gb1r, gb2r * gb2r) + sb;
      }
      ss += q == 0 ? sb : 0.f;
    }
    ss = wave_sum_fast(ss);
    if (lane == 0 && owner) {
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
      if (xcd_local) __hip_atomic_store(nx + (step & 1) * 32 + role * 8 + w, ((u64)tag << 32) | (u64)__float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(nx + (step & 1) * 32 + role * 8 + w, ((u64)tag << 32) | (u64)__float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // this workgroup reads its OWN eight partials from LDS (same floats, same summation order as everybody else's view of them)
      if (!OWNER) {      // (OWNER: a role's eight granules come from four workgroups: all 24 are read from the shared area)
        sm[S::MISC + 24 + role * 8 + w] = ss;
        if (book && role == 0) sm[S::MISC + 12] = want_stop ? 1.f : 0.f;
      }
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
          if (last_mb && (OWNER || half == 0)) { float* stats = KARGS()->stats; stats[32 + epoch] = mean_kl; stats[7] = mean_kl; }
        } else {
          const float vl = mb_s0 * inv_nb;
          lds_add(acc_vl, vl);
          *acc_last = vl;
        }
      }
    }
    STAMP(4)   // exchange + gradient norm + publish
    constexpr bool APRE = ICRL_HALVES_ADAM_PRELOAD && !OWNER;
    f32x4 ap_w1 = f32x4{0.f, 0.f, 0.f, 0.f}, ap_w2a = ap_w1, ap_w2b = ap_w1, ap_wh = ap_w1, ap_b = ap_w1;
    if (APRE) {
      ap_w1 = load_own_w1(); ap_w2a = load_own_w2(0); ap_w2b = load_own_w2(1);
      if (lowk) { ap_wh = load_own_wh(); ap_b = f32x4{sm[S::B1 + jb], sm[S::B2 + jb], sm[ex_s], 0.f}; }
    }
    // ---- while the granules travel: stage the next minibatch (rows -> the other X^T buffer, advantage statistics)
    const int xnext = xcur == S::XT0 ? S::XT1 : S::XT0;
    const int nb_next = __builtin_amdgcn_readfirstlane(ps_next.z) & NB_MASK;
    const int ptid = tid - 64 * ((NQ == 4 && !ICRL_HALVES_NORM_LOOK_EARLY) ? ICRL_HALVES_POLL_WAVE : 0);      // lane of the polling wave <-> granule
    const bool poller = ptid >= 0 && ptid < 24 && (OWNER || (ptid >> 3) != role);
    const u64* const slot = nx + (step & 1) * 32 + ((ptid >= 0 && ptid < 24) ? ptid : 0);
#if ICRL_HALVES_EARLY_POLL
    u64 v_first = 0;
#if ICRL_HALVES_NORM_LOOK_EARLY
    if (NQ == 4 && !OWNER) v_first = v_early;
    if (poller && (unsigned)((v_first >> 32) & 0x7fffffffu) != step) v_first = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    if (poller) v_first = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#endif
    if (!STAGE_HOP) {
      commit_rows(xnext);
      stats_partials(nb_next);
    }
    xcur = xnext;
    if (poller) {
      u64 v = 0;
      int spins = 0;
      bool ok = false;
#if ICRL_HALVES_EARLY_POLL
      v = v_first;
      ok = (unsigned)((v >> 32) & 0x7fffffffu) == step;
#endif
      while (!ok && spins < (1 << 24)) {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)((v >> 32) & 0x7fffffffu) == step) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
        ++spins;
      }
      sm[S::MISC + 24 + ptid] = __uint_as_float((unsigned)(v & 0xffffffffu));
      if (ptid == 7) sm[S::MISC + 12] = (v >> 63) ? 1.f : 0.f;     // the granule of the policy's book-keeping wave carries the stop flag
      if (!ok) sm[S::MISC + 13] = 1.f;
    }
    lds_barrier();   // (S6) norm partials, next minibatch and its statistics visible
    STAMP(5)   // staging + granule wait
    if (deep_pf) {      // the rows of step st + 2 (g_chunk = st + 1 here; this step's hop has just committed those of st + 1)
      const int idx_now = idx_next;
      idx_next = idx_nx2;
      idx_nx2 = chunk_idx(pc_nx3);
      pc_nx3 = ld_chunk(g_chunk + 4);
      issue_rows(idx_now);
    }
    float total = 0.f;
    {
#pragma unroll
      for (int g = 0; g < 6; ++g) {     // fixed order: every wave of every role and half forms the same total
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
      // (ICRL_HALVES_ADAM_PRE: m and v arrive already scaled by beta1 / beta2 — same products, formed earlier)
      auto adamn = [&](auto NE, const f32x4& g, f32x4& m, f32x4& v, f32x4& p) {   // stage by stage: NE independent chains (elements 0 .. NE - 1)
        constexpr int E = decltype(NE)::value;
        f32x4 d = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < E; ++i) {
          m[i] = fmaf(cw1, g[i], ICRL_HALVES_ADAM_PRE ? m[i] : omw1 * m[i]);
          v[i] = fmaf(c2w2, g[i] * g[i], ICRL_HALVES_ADAM_PRE ? v[i] : b2f_ * v[i]);
        }
#pragma unroll
        for (int i = 0; i < E; ++i) d[i] = fmaf(__builtin_amdgcn_sqrtf(v[i]), inv_bc2_sqrt, epsf);   // v_sqrt_f32 / v_rcp_f32: 1 ulp each
#pragma unroll
        for (int i = 0; i < E; ++i) p[i] = fmaf(-step_size, m[i] * __builtin_amdgcn_rcpf(d[i]), p[i]);
      };
      auto adam4 = [&](const f32x4& g, f32x4& m, f32x4& v, f32x4& p) { adamn(std::integral_constant<int, 4>{}, g, m, v, p); };
      // pad elements (k >= obs, o >= n_out) have g = m = v = p = 0 and stay 0: no masks needed
      if constexpr (OWNER) {
        // ---- one part per wave: the owner runs Adam and sends the updated parameters to the same wave of the other three parts
        const int h3 = H3_BASE + (((int)(step & 1) * 3 + role) * 4 + (w >> 1)) * H3_BLK;      // the owner's block of this step parity
        f32x4 pW1, pW2a, pW2b, pWh = f32x4{0.f, 0.f, 0.f, 0.f}, pB = pWh;
        if (owner) {
          pW1 = load_own_w1(); adam4(gW1r, mW1, vW1, pW1);
          pW2a = load_own_w2(0); adam4(gW2r[0], mW2[0], vW2[0], pW2a);
          pW2b = load_own_w2(1); adam4(gW2r[1], mW2[1], vW2[1], pW2b);
          raw_store(h3 + (0 * THH + tid) * 16, pW1); raw_store(h3 + (1 * THH + tid) * 16, pW2a); raw_store(h3 + (2 * THH + tid) * 16, pW2b);
          if (lowk) {
            f32x4 g_ = f32x4{gb1r, gb2r, ex_g >= 0 ? gex : 0.f, 0.f}, m_ = f32x4{mb1, mb2, mex, 0.f}, v_ = f32x4{vb1, vb2, vex, 0.f};
            pB = f32x4{sm[S::B1 + jb], sm[S::B2 + jb], sm[ex_s], 0.f};
            adam4(g_, m_, v_, pB);
            mb1 = m_[0]; mb2 = m_[1]; mex = m_[2]; vb1 = v_[0]; vb2 = v_[1]; vex = v_[2];
            pWh = load_own_wh(); adam4(gWhr, mWh, vWh, pWh);
            raw_store(h3 + (3 * THH + tid) * 16, pWh); raw_store(h3 + (4 * THH + tid) * 16, pB);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // acknowledged by the L2 the others read through -> this wave's flag
          if (lane == 0) {
            if (xcd_local) __builtin_amdgcn_raw_buffer_store_b32(step, grs, h3 + H3_FLAG + 64 * w, 0, 1);
            else __builtin_amdgcn_raw_buffer_store_b32(step, grs, h3 + H3_FLAG + 64 * w, 0, 16);
          }
        } else {
          int spins = 0;
          while (true) {
            const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(grs, h3 + H3_FLAG + 64 * w, 0, 16);
            if (f == step) break;
            if (++spins >= (1 << 22)) { sm[S::MISC + 13] = 1.f; status = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          asm volatile("" ::: "memory");
          pW1 = raw_load(h3 + (0 * THH + tid) * 16); pW2a = raw_load(h3 + (1 * THH + tid) * 16); pW2b = raw_load(h3 + (2 * THH + tid) * 16);
          if (lowk) { pWh = raw_load(h3 + (3 * THH + tid) * 16); pB = raw_load(h3 + (4 * THH + tid) * 16); }
        }
        store_w1(pW1); store_w2(0, pW2a); store_w2(1, pW2b);
        if (lowk) {
          if (q == 0) { sm[S::B1 + jb] = pB[0]; sm[S::B2 + jb] = pB[1]; sm[ex_s] = pB[2]; }
          store_wh(pWh);
          __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the log_std store has landed before refresh_gauss re-reads it
          refresh_gauss();
        }
      } else {
      { f32x4 p_ = APRE ? ap_w1 : load_own_w1(); adam4(gW1r, mW1, vW1, p_); store_w1(p_); }
      if (lowk) {
        f32x4 g_ = f32x4{gb1r, gb2r, ex_g >= 0 ? gex : 0.f, 0.f}, p_ = APRE ? ap_b : f32x4{sm[S::B1 + jb], sm[S::B2 + jb], sm[ex_s], 0.f};
        f32x4 m_ = f32x4{mb1, mb2, mex, 0.f}, v_ = f32x4{vb1, vb2, vex, 0.f};
        // identical arithmetic in the four q lanes, lane q == 0 stores; the third element exists on the head-bias / log_std waves only
        const bool has_ex = w == W_BH || (!DISC && role == 0 && w == W_LS);
        if (!ICRL_HALVES_ADAM_TRIM) adam4(g_, m_, v_, p_);
        else if (has_ex) adamn(std::integral_constant<int, 3>{}, g_, m_, v_, p_);
        else adamn(std::integral_constant<int, 2>{}, g_, m_, v_, p_);
        mb1 = m_[0]; mb2 = m_[1]; mex = m_[2]; vb1 = v_[0]; vb2 = v_[1]; vex = v_[2];
        if (q == 0) { sm[S::B1 + jb] = p_[0]; sm[S::B2 + jb] = p_[1]; sm[ex_s] = p_[2]; }
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) { f32x4 p_ = APRE ? (cc == 0 ? ap_w2a : ap_w2b) : load_own_w2(cc); adam4(gW2r[cc], mW2[cc], vW2[cc], p_); store_w2(cc, p_); }
      if (lowk) {
        {   // head weights: element i of a lane is output 4 i + q — only the k groups that hold an output can hold a parameter
          f32x4 p_ = APRE ? ap_wh : load_own_wh();
          const int ngw = (n_out + 3) >> 2;
          if (!ICRL_HALVES_ADAM_TRIM || ngw > 2) adam4(gWhr, mWh, vWh, p_);
          else if (ngw == 2) adamn(std::integral_constant<int, 2>{}, gWhr, mWh, vWh, p_);
          else adamn(std::integral_constant<int, 1>{}, gWhr, mWh, vWh, p_);
          store_wh(p_);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the log_std store has landed before refresh_gauss re-reads it
        refresh_gauss();
      }
      }      // replicated Adam
    }
    lds_barrier();   // (S7) updated weights visible
    STAMP(6)   // Adam
  }  // optimiser steps

  __syncthreads();
  // ---- write back weights, moments, statistics: the parts are replicas, part 0 writes (OWNER: every wave's owner — the moments live there only)
  if (OWNER ? owner : half == 0) {
  const TrainArgs* kw = ka;
  asm volatile("" : "+s"(kw));
  const TrainArgs& a = *kw;
  const PolLayout& L = a.L;
  const int gW1 = L.W1[role], gb1 = L.b1[role], gW2 = L.W2[role], gb2 = L.b2[role];
  const int gWh = role == 0 ? L.Wa : (role == 1 ? L.Wv : L.Wc);
  {
    const f32x4 pv = load_own_w1();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * jt + 4 * q + i, k = 16 * kh + r;
      if (k < O) { a.params[gW1 + j * O + k] = pv[i]; a.exp_avg[gW1 + j * O + k] = mW1[i]; a.exp_avg_sq[gW1 + j * O + k] = vW1[i]; }
    }
  }
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const f32x4 pv = load_own_w2(cc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * jt + 4 * q + i, k = 16 * (2 * kh + cc) + r;
      a.params[gW2 + j * HD + k] = pv[i];
      a.exp_avg[gW2 + j * HD + k] = mW2[cc][i];
      a.exp_avg_sq[gW2 + j * HD + k] = vW2[cc][i];
    }
  }
  if (lowk) {
    const f32x4 pv = load_own_wh();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = out_of(i), j = 16 * jt + r;
      if (o < n_out) { a.params[gWh + o * HD + j] = pv[i]; a.exp_avg[gWh + o * HD + j] = mWh[i]; a.exp_avg_sq[gWh + o * HD + j] = vWh[i]; }
    }
    if (q == 0 && ex_g >= 0) { a.params[ex_g] = sm[ex_s]; a.exp_avg[ex_g] = mex; a.exp_avg_sq[ex_g] = vex; }
    if (q == 0) {
      a.params[gb1 + jb] = sm[S::B1 + jb]; a.exp_avg[gb1 + jb] = mb1; a.exp_avg_sq[gb1 + jb] = vb1;
      a.params[gb2 + jb] = sm[S::B2 + jb]; a.exp_avg[gb2 + jb] = mb2; a.exp_avg_sq[gb2 + jb] = vb2;
    }
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

template <int NT1, bool DISC, int OBS, int NQ, bool PROF>
__global__ void __launch_bounds__(THH) ppo_train_halves_kernel(TrainArgs a, int packed) {
  int run = 0, j = (int)blockIdx.x;
  if (packed && !packed_slot(3 * NQ, 1, run, j)) return;
  const TrainArgs* const ka = (const TrainArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (ICRL_HALVES_ROLE_SPEC && OBS == 18 && NQ == 4 && !DISC) {      // HCWithPos (BASELINE configs[1], [3]): one body per role
    const int role = j % 3;
    if (role == 0) ppo_train_halves_body<NT1, DISC, OBS, false, NQ, 0, PROF>(a, ka, j);
    else if (role == 1) ppo_train_halves_body<NT1, DISC, OBS, false, NQ, 1, PROF>(a, ka, j);
    else ppo_train_halves_body<NT1, DISC, OBS, false, NQ, 2, PROF>(a, ka, j);
  } else {
    ppo_train_halves_body<NT1, DISC, OBS, false, NQ, -1, PROF>(a, ka, j);
  }
}

// several independent runs in ONE launch: the packed 1-D grid of ppo_common.h (a run's workgroups on one XCD), or grid (3 NQ, n_runs)
// with run = blockIdx.y when that many workgroups are not resident on their XCDs at once; the argument blocks live in device memory
template <int NT1, bool DISC, int OBS, int NQ, bool PROF>
__global__ void __launch_bounds__(THH) ppo_train_halves_batch_kernel(const TrainArgs* __restrict__ runs, int n_runs, int packed) {
  int run = (int)blockIdx.y, j = (int)blockIdx.x;
  if (packed && !packed_slot(3 * NQ, n_runs, run, j)) return;
  const TrainArgs* const ka = as_global(runs + run);
  ppo_train_halves_body<NT1, DISC, OBS, true, NQ, -1, PROF>(*ka, ka, j);
}

template <int NT1, bool DISC, int OBS, int NQ, bool PROF>
static int launch_halves(const TrainArgs* one, const TrainArgs* d_args, int n_runs, hipStream_t s) {
  static_assert(SmemH<NT1>::TOTAL * sizeof(float) <= 160 * 1024, "LDS budget");
  if (one != nullptr) {
    TrainArgs arg = *one;
    return launch_update_single(ppo_train_halves_kernel<NT1, DISC, OBS, NQ, PROF>, 3 * NQ, dim3(THH), 0, s, arg);
  }
  const int pg = packed_grid(3 * NQ, n_runs);
  hipLaunchKernelGGL((ppo_train_halves_batch_kernel<NT1, DISC, OBS, NQ, PROF>), pg ? dim3(pg) : dim3(3 * NQ, n_runs), dim3(THH), 0, s, d_args, n_runs, pg ? 1 : 0);
  return (int)hipGetLastError();
}

template <int NQ, bool PROF>
static int dispatch_halves_p(const TrainArgs* one, const TrainArgs* d_args, int n_runs, int obs, bool discrete, hipStream_t s) {
  if (!discrete && obs == 18) return launch_halves<2, false, 18, NQ, PROF>(one, d_args, n_runs, s);      // HCWithPos (BASELINE configs[1], [3])
  if (discrete && obs == 1) return launch_halves<2, true, 1, NQ, PROF>(one, d_args, n_runs, s);          // LapGridWorld (configs[0])
  return discrete ? launch_halves<2, true, 0, NQ, PROF>(one, d_args, n_runs, s) : launch_halves<2, false, 0, NQ, PROF>(one, d_args, n_runs, s);
}
// prof: hp._pad & 1 of the run(s) — the instantiation with the phase timers (tools only)
template <int NQ>
static int dispatch_halves(const TrainArgs* one, const TrainArgs* d_args, int n_runs, int obs, bool discrete, bool prof, hipStream_t s) {
  return prof ? dispatch_halves_p<NQ, true>(one, d_args, n_runs, obs, discrete, s) : dispatch_halves_p<NQ, false>(one, d_args, n_runs, obs, discrete, s);
}

// obs <= 32 (nt1 <= 2), a.gx set and zeroed (prepare_train); parts = 2 | 4 workgroups per network
int launch_train_halves(const TrainArgs& a, bool discrete, int parts, hipStream_t s) {
  const bool prof = (a.hp._pad & 1) != 0;
  return parts == 4 ? dispatch_halves<4>(&a, nullptr, 1, a.L.O, discrete, prof, s) : dispatch_halves<2>(&a, nullptr, 1, a.L.O, discrete, prof, s);
}
int launch_train_halves_batch(const TrainArgs* d_args, int n_runs, int obs, bool discrete, int parts, bool prof, hipStream_t s) {
  return parts == 4 ? dispatch_halves<4>(nullptr, d_args, n_runs, obs, discrete, prof, s) : dispatch_halves<2>(nullptr, d_args, n_runs, obs, discrete, prof, s);
}

}  // namespace icrl
