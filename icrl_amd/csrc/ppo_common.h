// Shared pieces of the two PPO-Lagrangian update kernels (ppo_train.hip: column-split tiles, any obs width;
// ppo_train_rows.hip: row-owning waves, obs <= 64) — gfx950.
#pragma once
#include "common.h"

namespace icrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int RB = 64;   // minibatch rows per chunk
constexpr int HD = 64;   // hidden width (both layers)
constexpr int SH = 72;   // LDS row stride of 64-wide matrices (= 8 mod 16: conflict-free ds_read_b128 operand fetch)
constexpr int MAXB = 256;   // minibatch rows: up to four 64-row chunks

struct TrainArgs {
  PolLayout L;
  float* params;
  float* exp_avg;
  float* exp_avg_sq;
  int* adam_t;
  icrl_buffer_t buf;
  const int* perms;
  const float* nu;
  icrl_ppo_hyper_t hp;
  float* stats;
  u64* xch;
  unsigned t_magic;   // floor(2^32 / T): fast division of a flat index by T
  const struct PlanStep* plan_steps;     // rows kernel: per optimiser step (+2 zero entries)
  const struct PlanChunk* plan_chunks;   // rows kernel: per 64-row chunk (+5 zero entries); two entries per step in split mode
  int n_steps;
  u64* gx;                               // split mode (two workgroups per network): partial-gradient granules, ICRL_PPO_SPLIT_BYTES
};

// the schedule of a launch, tabulated once by ppo_plan_kernel so that the persistent kernel carries no epoch / minibatch /
// cursor arithmetic in scalar registers: which rows of which permutation form each minibatch and chunk, and Adam's bias
// corrections (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), evaluated in double like torch)
struct PlanStep {
  float step_size, inv_bc2_sqrt;
  int nb_flags;    // rows (10 bits) | first minibatch of the epoch << 10 | last << 11 | epoch << 12

  int perm_base;   // epoch * T*N + position of the minibatch's first row in the permutation
};
constexpr int NB_MASK = 0x3ff, NB_FIRST = 10, NB_LAST = 11, NB_EPOCH = 12;
struct PlanChunk {
  int perm_base, rows;
};

// sum over the 16 lanes sharing lane/16
__device__ __forceinline__ float sum16(float v) { return row_sum(v); }   // DPP row operations (common.h)

// column sums of a wave's 16x16 accumulator tile: every lane ends with the sum over the tile's 16 rows of column lane%16
__device__ __forceinline__ float tile_colsum(const f32x4& t) {
  return quad_rows_sum((t[0] + t[1]) + (t[2] + t[3]));
}

// diagnostic phase timer (only when hp._pad != 0: the stamp drains the LDS queue, so never in a timed run)
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define STAMP(slot)                                                        \
  if (prof) {                                                              \
    __builtin_amdgcn_sched_barrier(0);                                     \
    const unsigned long long now_ = stamp();                               \
    __builtin_amdgcn_sched_barrier(0);                                     \
    ph[slot] += now_ - t_last;                                             \
    t_last = now_;                                                         \
  }

// workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the global loads of the row prefetch
// (s_waitcnt vmcnt(0)) at every one of the ~13 barriers of a step and expose their full latency each time
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x4 lds128(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// a value parked in the accumulation-register half of the (unified, 512-entry) register file: VALU instructions cannot name
// those registers, so the value costs nothing of the 256 VALU-addressable ones while it is not in use.  Both directions are one
// v_accvgpr move.  (The "a" constraint makes the register allocator keep the value in an AGPR between the two asm statements.)
__device__ __forceinline__ void acc_put(float& slot, float v) { asm("v_accvgpr_write_b32 %0, %1" : "=a"(slot) : "v"(v)); }
// in-place update of a loop-carried slot: the tied operand pins the value to ONE register across the loop (with a fresh "=a"
// definition per iteration the allocator re-homes the slot and copies it back on the loop edge, one v_accvgpr_mov per value)
__device__ __forceinline__ void acc_set(float& slot, float v) { asm("v_accvgpr_write_b32 %0, %1" : "+a"(slot) : "v"(v)); }
__device__ __forceinline__ float acc_get(const float& slot) {
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(slot));
  return v;
}

// cursor into the stream of minibatch rows: epoch, position inside the epoch, position inside the minibatch
struct Cursor {
  int e, p, m;
};

// ---------------------------------------------------------------------------------------------------------------
// XCD placement of the persistent update workgroups
// ---------------------------------------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs, so workgroups b and b + 8 of a 1-D grid share one.  The update launches use that:
// the M workgroups of run r sit at b = 8 M (r / 8) + (r % 8) + 8 j, j = 0 .. M-1 (the other workgroups of the grid leave at once), and a
// granule that one of them stores with workgroup scope (`sc0`) STAYS in that XCD's L2, where the others' agent-scope (`sc1`: L1
// bypassed, L2-served) polls find it; an `sc1` store drops the line and every reader pays the trip through the fabric.  Measured on
// MI355X: HC update 8.50 -> 8.31 us per step (three norm granules per step), AntWall B = 128 22.2 -> 20.4 (28 k gradient granules).
// Placement is a speed matter only: every workgroup publishes its XCD (HW_REG_XCC_ID) once per launch in a spare exchange word and
// uses the `sc0` stores only when all workgroups of its run reported the same one; any other dispatch keeps the agent-scope stores.
// Every granule carries its step tag, so a poll that is served a stale line retries.
constexpr int XCD_STRIDE = 8;
// spare words of the 64-word norm exchange area (both kernels' layouts leave 28..31 and 60..63 unused; zeroed before every launch)
__device__ __forceinline__ int xcc_word(int j) { return j < 4 ? 28 + j : 56 + j; }
// the wave-quad kernel with FOUR workgroups per network (12 per run): its norm granules use words 0..23 and 32..55 of the area, 24..31 and 56..63 are free
__device__ __forceinline__ int xcc_word12(int j) { return j < 8 ? 24 + j : 48 + j; }
// called by ONE thread of workgroup j (of M <= 6; M <= 12 with `wide`) of a run: true when all M report the same XCD (common.h: all_on_one_xcd)
__device__ __forceinline__ bool run_on_one_xcd(unsigned long long* xch, int j, int M, bool wide = false) {
  const unsigned long long me = 0x100ull | (unsigned long long)xcc_id();
  __hip_atomic_store(xch + (wide ? xcc_word12(j) : xcc_word(j)), me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool same = true;
  for (int o = 0; o < M; ++o) {
    if (o == j) continue;
    unsigned long long v = 0;
    for (int spins = 0; spins < (1 << 18); ++spins) {
      v = __hip_atomic_load(xch + (wide ? xcc_word12(o) : xcc_word(o)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
    same = same && v == me;
  }
  return same;
}
// blockIdx.x -> (run, j) of the packed layout; false: this workgroup has nothing to do
__device__ __forceinline__ bool packed_slot(int M, int n_runs, int& run, int& j) {
  const int id = (int)blockIdx.x, per = XCD_STRIDE * M;
  run = (id / per) * XCD_STRIDE + (id & (XCD_STRIDE - 1));
  j = (id % per) / XCD_STRIDE;
  return run < n_runs;
}
// grid of the packed layout; 0: the runs' workgroups could not all be resident on their XCDs at once (one workgroup per CU, two CUs
// of every XCD left to whatever else is in flight — the side-stream permutation sorts of a seed batch are short kernels that come and
// go: they can delay a workgroup's dispatch, they cannot hold a CU against it, and the spins are bounded in seconds) — the caller then
// uses the run-major layout, whose runs become resident oldest first
inline int packed_grid(int M, int n_runs) {
  static const bool off = getenv("ICRL_NO_XCD_PACK") != nullptr;      // tests / A/B: the run-major layout (agent-scope stores) everywhere
  static const int cus_per_xcd = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < XCD_STRIDE) return 32;
    return cus / XCD_STRIDE;
  }();
  const int groups = (n_runs + XCD_STRIDE - 1) / XCD_STRIDE;
  if (off || groups * M > cus_per_xcd - 2) return 0;
  return n_runs >= XCD_STRIDE ? groups * XCD_STRIDE * M : XCD_STRIDE * (M - 1) + n_runs;
}
// a single-run update launch: the packed 1-D grid as a cooperative launch; a device or partition that refuses it
// (hipErrorCooperativeLaunchTooLarge: 8 (M - 1) + 1 workgroups of which M work) gets the plain M-workgroup grid
template <class K>
inline int launch_update_single(K kernel, int M, dim3 block, size_t dyn_lds, hipStream_t s, TrainArgs& arg) {
  const int pg = packed_grid(M, 1);
  hipError_t e = launch_coresident(kernel, dim3(pg ? pg : M), block, dyn_lds, s, arg, pg ? 1 : 0);
  if (e == hipErrorCooperativeLaunchTooLarge && pg) {
    (void)hipGetLastError();
    e = launch_coresident(kernel, dim3(M), block, dyn_lds, s, arg, 0);
  }
  return (int)e;
}

// The `sc0` granule / raw stores rest on gfx942 / gfx950 behaviour (write-through L1, one L2 per XCD that every CU of the XCD reads
// through, checked per launch by run_on_one_xcd): not a property of the HIP memory model.  This library is built for gfx950 only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "libicrl_hip: the XCD-local exchange of the persistent kernels is written for gfx950 (CDNA4) / gfx942"
#endif

// ppo_train_rows.hip: row-owning-wave kernel (nt1 = ceil(obs / 16) <= 8), one wave per SIMD
int launch_train_rows(const TrainArgs& a, int nt1, bool discrete, bool split, hipStream_t s);
// ppo_train_pairs.hip: wave-pair kernel, two waves per SIMD (nt1 rounded up to an even tile count)
int launch_train_pairs(const TrainArgs& a, int nt1, bool discrete, hipStream_t s);
// ppo_train_halves.hip: two workgroups per network, wave quads (nt1 <= 2; single-run launches; a.gx set)
// parts = 2 | 4 workgroups per network (round 6: four — 16 rows of every chunk each)
int launch_train_halves(const TrainArgs& a, bool discrete, int parts, hipStream_t s);
int launch_train_halves_batch(const TrainArgs* d_args, int n_runs, int obs, bool discrete, int parts, bool prof, hipStream_t s);      // prof: hp._pad & 1 (the runs of a batch share it)
// ppo_train_quarters.hip: four workgroups per network at obs 65..128 (wave quads, the row-owning kernel's parameter ownership); single-run launches
int launch_train_quarters_wide(const TrainArgs& a, bool discrete, hipStream_t s);
// ppo_train_quarters2.hip: the same for minibatches of 65..128 rows: both chunks in one pass, two row tiles per wave (chunk plan with two entries per step)
int launch_train_quarters_wide2(const TrainArgs& a, bool discrete, hipStream_t s);
constexpr int HALVES_MAX_RUNS = 40;      // batched launches: 6 workgroups per run, 5 groups of 8 runs = 30 workgroups per XCD (packed_grid)
constexpr int QUARTERS_MAX_RUNS = 16;    // 12 workgroups per run, 2 groups of 8 runs = 24 workgroups per XCD
// batched forms: n_runs argument blocks in DEVICE memory, grid.y = run
int launch_train_rows_batch(const TrainArgs* d_args, int n_runs, int nt1, bool discrete, bool split, hipStream_t s);
int launch_train_pairs_batch(const TrainArgs* d_args, int n_runs, int obs, int nt1, bool discrete, bool prof, hipStream_t s);


}  // namespace icrl
