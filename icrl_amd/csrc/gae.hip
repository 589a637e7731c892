// Dual (reward + cost) GAE scan over a time-major [T, N] rollout — gfx950.
//
// Replaces RolloutBufferWithCost._compute_returns_and_advantage x2
// (ref: stable_baselines3/common/buffers.py:493-552) with ONE launch that reads r, c, V_r, V_c, dones once
// and writes A_r, A_c, R_r, R_c once: 36 B per transition, HBM-bound.
//
// Layout / mapping
//   * lane <-> environment column: a wave's 64 lanes read 64 consecutive floats of one time row (256 B,
//     fully coalesced); time runs in the register dimension.
//   * the recurrence A_t = delta_t + c_t * A_{t+1} is carried in float64 registers exactly as the reference's
//     numpy code ends up doing (bool `dones` promote the accumulator to float64; delta_t itself is float32
//     arithmetic for t < T-1).  Loads/stores stay float32.
//   * large N (>= 512 column tiles): one wave per tile walks all T rows with a double-buffered batch of U rows
//     in flight (5*U independent 256-B loads per wave) — bit-identical to the sequential reference.
//   * small N (the BASELINE configs: 1..8 tiles): W waves of a workgroup split the time axis.  Pass 1 reduces
//     each chunk to its affine map (P, Q) with A_in = Q + P * A_out (associative scan over affine maps,
//     SURVEY.md §5 "long-context" row), the maps are exchanged through LDS, every wave folds the maps of the
//     later chunks into its carry-in and pass 2 replays its chunk (inputs now L2-resident) writing outputs.
//     The fold re-associates float64 products, so small-N results can differ from the sequential scan in the
//     last float64 bit; after the float32 store that is invisible except for ~1e-8 of elements (<= 1 ulp).
//
// Built with -ffp-contract=off: every multiply/add below rounds separately, like the numpy reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "common.h"

using icrl::fail;

namespace {

constexpr int U_DEFAULT = 8;  // time rows per register batch (x2 for the double buffer)

struct GaeArgs {
  const float *r, *c, *vr, *vc, *d, *lvr, *lvc;
  const uint8_t* ld;
  float *ar, *ac, *rr, *rc;
  int T, N;
  float g_r, gl_r, g_c, gl_c;
};

template <int U>
struct Batch {
  float r[U], c[U], vr[U], vc[U], d[U];
};

template <bool NT>
__device__ __forceinline__ float ldg(const float* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void stg(float* p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int U, bool NT>
__device__ __forceinline__ void load_batch(const GaeArgs& a, unsigned n, int t_top, int t0, Batch<U>& b) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int t = t_top - u;
    t = t < t0 ? t0 : t;  // clamped (uniform) — the value is ignored below t0
    const size_t row = (size_t)t * a.N;  // wave-uniform: scalar row base + 32-bit lane offset
    b.r[u] = ldg<NT>(a.r + row + n);
    b.c[u] = ldg<NT>(a.c + row + n);
    b.vr[u] = ldg<NT>(a.vr + row + n);
    b.vc[u] = ldg<NT>(a.vc + row + n);
    b.d[u] = ldg<NT>(a.d + row + n);
  }
}

// carried state of one chunk walk
struct Carry {
  double Ar, Ac;      // running advantages (or Q of the affine map in the composite pass)
  double Pr, Pc;      // product of coefficients (composite pass only)
  float vr_next, vc_next, d_next;
};

template <bool WRITE, int U, bool NT>
__device__ __forceinline__ void run_batch(const GaeArgs& a, unsigned n, bool live, int t_top, int t0, const Batch<U>& b, Carry& s) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int t = t_top - u;
    if (t >= t0) {
      const float nnt = 1.0f - s.d_next;
      // ref: buffers.py:535-537 — float32 delta, float32 coefficient, float64 accumulate
      const float gvr = (a.g_r * s.vr_next) * nnt;
      const float gvc = (a.g_c * s.vc_next) * nnt;
      const float dr = (b.r[u] + gvr) - b.vr[u];
      const float dc = (b.c[u] + gvc) - b.vc[u];
      const float cr = a.gl_r * nnt;
      const float cc = a.gl_c * nnt;
      s.Ar = (double)dr + (double)cr * s.Ar;
      s.Ac = (double)dc + (double)cc * s.Ac;
      if (WRITE) {
        if (live) {
          const size_t row = (size_t)t * a.N;
          const float fr = (float)s.Ar, fc = (float)s.Ac;
          stg<NT>(a.ar + row + n, fr);
          stg<NT>(a.ac + row + n, fc);
          stg<NT>(a.rr + row + n, fr + b.vr[u]);
          stg<NT>(a.rc + row + n, fc + b.vc[u]);
        }
      } else {
        s.Pr = (double)cr * s.Pr;
        s.Pc = (double)cc * s.Pc;
      }
      s.vr_next = b.vr[u];
      s.vc_next = b.vc[u];
      s.d_next = b.d[u];
    }
  }
}

// Walk t = t1-1 .. t0 for column n.  WRITE=false: compute the chunk's affine map into (s.Pr,s.Ar),(s.Pc,s.Ac)
// starting from A=0,P=1.  WRITE=true: s.Ar/s.Ac hold the carry-in and outputs are stored.
template <bool WRITE, int U, bool NT>
__device__ __forceinline__ void walk_chunk(const GaeArgs& a, unsigned n, bool live, int t0, int t1, Carry& s) {
  int t = t1 - 1;
  if (t1 == a.T) {
    // t = T-1: bootstrap from last values; `1.0 - last_dones(bool)` is float64 in the reference (buffers.py:530-531)
    const size_t off = (size_t)t * a.N + n;
    const float rew = a.r[off], cost = a.c[off], vr = a.vr[off], vc = a.vc[off], d = a.d[off];
    const double nnt = 1.0 - (a.ld[n] ? 1.0 : 0.0);
    const float gvr = a.g_r * a.lvr[n];
    const float gvc = a.g_c * a.lvc[n];
    s.Ar = ((double)rew + (double)gvr * nnt) - (double)vr;  // + gamma*lambda*nnt*0
    s.Ac = ((double)cost + (double)gvc * nnt) - (double)vc;
    if (WRITE) {
      if (live) {
        const float fr = (float)s.Ar, fc = (float)s.Ac;
        a.ar[off] = fr;
        a.ac[off] = fc;
        a.rr[off] = fr + vr;
        a.rc[off] = fc + vc;
      }
    } else {
      s.Pr = 0.0;  // nothing beyond T feeds in
      s.Pc = 0.0;
    }
    s.vr_next = vr;
    s.vc_next = vc;
    s.d_next = d;
    --t;
  } else {
    const size_t off = (size_t)t1 * a.N + n;
    s.vr_next = a.vr[off];
    s.vc_next = a.vc[off];
    s.d_next = a.d[off];
  }
  if (t < t0) return;
  Batch<U> b0, b1;
  load_batch<U, NT>(a, n, t, t0, b0);
  while (true) {
    if (t - U >= t0) load_batch<U, NT>(a, n, t - U, t0, b1);
    run_batch<WRITE, U, NT>(a, n, live, t, t0, b0, s);
    t -= U;
    if (t < t0) break;
    if (t - U >= t0) load_batch<U, NT>(a, n, t - U, t0, b0);
    run_batch<WRITE, U, NT>(a, n, live, t, t0, b1, s);
    t -= U;
    if (t < t0) break;
  }
}

template <int W, int U, bool NT, int G = 1>
__global__ void __launch_bounds__(64 * W * G) gae_dual_kernel(GaeArgs a) {
  const int lane = threadIdx.x & 63;
  static_assert(W == 1 || G == 1, "G independent waves per workgroup only for the one-wave-per-tile shape");
  const int wave = G > 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> SGPR
  const int col = G > 1 ? blockIdx.x * 64 * G + threadIdx.x : blockIdx.x * 64 + lane;
  const bool live = col < a.N;
  const unsigned n = live ? col : a.N - 1;
  const int chunk = (a.T + W - 1) / W;
  const int t0 = wave * chunk;
  const int t1 = (t0 + chunk < a.T) ? t0 + chunk : a.T;
  Carry s;
  s.Ar = 0.0; s.Ac = 0.0; s.Pr = 1.0; s.Pc = 1.0;
  s.vr_next = 0.f; s.vc_next = 0.f; s.d_next = 0.f;
  if (W > 1) {
    __shared__ double maps[W][4][64];
    if (t0 < t1) walk_chunk<false, U, NT>(a, n, live, t0, t1, s);
    maps[wave][0][lane] = s.Pr;
    maps[wave][1][lane] = s.Ar;
    maps[wave][2][lane] = s.Pc;
    maps[wave][3][lane] = s.Ac;
    __syncthreads();
    double Ar = 0.0, Ac = 0.0;
    for (int w = W - 1; w > wave; --w) {
      Ar = maps[w][1][lane] + maps[w][0][lane] * Ar;
      Ac = maps[w][3][lane] + maps[w][2][lane] * Ac;
    }
    s.Ar = Ar;
    s.Ac = Ac;
  }
  if (t0 < t1) walk_chunk<true, U, NT>(a, n, live, t0, t1, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// small N, long T (the BASELINE configs: 1..16 column tiles x 256..2048 rows): the time axis is split over C workgroups of
// SW waves each, so a wave walks T / (C * SW) rows instead of T / 16.  Level 1 is the scan above inside a workgroup; level 2
// composes each workgroup's SW maps into one, publishes it (2 KB + a flag holding this launch's tag) and folds the maps of the
// workgroups holding LATER rows into the carry-in.  A workgroup only ever waits for workgroups with a lower blockIdx (the
// latest rows are blockIdx 0), which the dispatcher has started before it: no co-residency assumption.
// Same re-association caveat as above (<= 1 ulp in float64 before the float32 store).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SPLIT_W = 8;         // waves per workgroup
constexpr int SPLIT_CMAX = 16;     // workgroups per column tile

constexpr unsigned GAE_SPIN_LIMIT = 1u << 22;   // polls of ~1 us each: seconds, never reached by a healthy launch

__device__ __forceinline__ void gae_dual_split_body(const GaeArgs& a, int C, unsigned tag, double* maps, unsigned* flags, unsigned* status = nullptr,
                                                    unsigned spin_limit = GAE_SPIN_LIMIT, int fault = 0) {
  constexpr int W = SPLIT_W, U = U_DEFAULT;
  __shared__ double own[W][4][64];
  __shared__ double ext[SPLIT_CMAX - 1][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x / C;
  const int c = C - 1 - (blockIdx.x - tile * C);       // time chunk of this workgroup: latest rows first
  const int col = tile * 64 + lane;
  const bool live = col < a.N;
  const unsigned n = live ? col : a.N - 1;
  const int Tc = (a.T + C - 1) / C;
  const int wt0 = c * Tc < a.T ? c * Tc : a.T, wt1 = wt0 + Tc < a.T ? wt0 + Tc : a.T;
  const int sub = (Tc + W - 1) / W;
  const int t0 = wt0 + wave * sub < wt1 ? wt0 + wave * sub : wt1;
  const int t1 = t0 + sub < wt1 ? t0 + sub : wt1;
  Carry s;
  s.Ar = 0.0; s.Ac = 0.0; s.Pr = 1.0; s.Pc = 1.0;
  s.vr_next = 0.f; s.vc_next = 0.f; s.d_next = 0.f;
  if (t0 < t1) walk_chunk<false, U, false>(a, n, live, t0, t1, s);
  own[wave][0][lane] = s.Pr;
  own[wave][1][lane] = s.Ar;
  own[wave][2][lane] = s.Pc;
  own[wave][3][lane] = s.Ac;
  __syncthreads();
  if (wave == 0 && c > 0 && !(fault && c == C - 1)) {  // the map of rows [wt0, wt1): needed by the chunks before it (fault: test injection, never published)
    double Pr = 1.0, Qr = 0.0, Pc = 1.0, Qc = 0.0;
    for (int w = W - 1; w >= 0; --w) {
      Qr = own[w][1][lane] + own[w][0][lane] * Qr;  Pr = own[w][0][lane] * Pr;
      Qc = own[w][3][lane] + own[w][2][lane] * Qc;  Pc = own[w][2][lane] * Pc;
    }
    double* m = maps + (size_t)(tile * C + c) * 256;
    m[lane] = Pr; m[64 + lane] = Qr; m[128 + lane] = Pc; m[192 + lane] = Qc;
    __hip_atomic_store(&flags[tile * C + c], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // wave-wide release: after every lane's stores
  }
  for (int j = c + 1 + wave; j < C; j += W) {
    unsigned polls = 0;                                // bounded like every other wait of the library (include/icrl_hip.h): a chunk that never
    while (__hip_atomic_load(&flags[tile * C + j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != tag) {   // arrives ends the wait with the status word set
      if (++polls >= spin_limit) { if (lane == 0 && status != nullptr) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    const double* m = maps + (size_t)(tile * C + j) * 256;
#pragma unroll
    for (int k = 0; k < 4; ++k) ext[j - c - 1][k][lane] = m[k * 64 + lane];
  }
  __syncthreads();
  double Ar = 0.0, Ac = 0.0;
  for (int j = C - 1; j > c; --j) {
    Ar = ext[j - c - 1][1][lane] + ext[j - c - 1][0][lane] * Ar;
    Ac = ext[j - c - 1][3][lane] + ext[j - c - 1][2][lane] * Ac;
  }
  for (int w = W - 1; w > wave; --w) {
    Ar = own[w][1][lane] + own[w][0][lane] * Ar;
    Ac = own[w][3][lane] + own[w][2][lane] * Ac;
  }
  s.Ar = Ar;
  s.Ac = Ac;
  if (t0 < t1) walk_chunk<true, U, false>(a, n, live, t0, t1, s);
}

__global__ void __launch_bounds__(64 * SPLIT_W) gae_dual_split_kernel(GaeArgs a, int C, unsigned tag, double* maps, unsigned* flags, unsigned* status,
                                                                      unsigned spin_limit, int fault) {
  gae_dual_split_body(a, C, tag, maps, flags, status, spin_limit, fault);
}

// ---------------------------------------------------------------------------------------------------------------------
// REGISTER-RESIDENT split scan (round 6): the mid range between the cache-resident loop sizes and the streaming shapes (and the loop
// sizes themselves).  One wave per tile has too few bytes in flight below ~65 536 envs (Little's law: 128 tiles x 20 KB against the
// ~10 MB 8 TB/s needs) and the two-pass split above re-reads its chunk.  Here a workgroup of 8 waves owns 128 rows of one column
// tile and every wave keeps its 16 rows x 5 arrays IN REGISTERS (80 VGPRs) between the two passes, so every byte is read ONCE:
//   load 16 rows (all 80 loads issued back to back) -> the chunk's affine map from registers -> level 1 through LDS -> wave 0
//   publishes the workgroup's map (write-through stores, `s_waitcnt vmcnt(0)`, one relaxed agent-scope flag: no release fence — a
//   `buffer_wbl2` per workgroup under a streaming grid costs microseconds) -> the maps of the LATER chunks are polled (bounded) and read
//   with agent-scope loads -> replay from registers, outputs stored once.
// grid = tiles x ceil(T / 128) workgroups (2 048 .. 16 384 for 8 192 .. 65 536 envs at T = 2048); a workgroup only waits for lower
// blockIdx (later rows), so the grid need not be co-resident.  Re-association as for the split scan above (<= 1 float32 ulp).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef ICRL_RS_R
#define ICRL_RS_R 16             // rows a wave holds in registers (A/B: 8 -> 64-row chunks, twice the workgroups and maps)
#endif
constexpr int RS_W = 8, RS_R = ICRL_RS_R, RS_TC = RS_W * RS_R;
constexpr int RS_CMAX = 2048 / RS_TC;      // T <= 2048

struct HeadRegs {
  float rew, cost, vr, vc, d, lvr, lvc;
  int ld;
};

template <bool NT>
__device__ __forceinline__ void rs_head_load(const GaeArgs& a, unsigned n, int t1, HeadRegs& h) {
  if (t1 == a.T) {
    const size_t off = (size_t)(a.T - 1) * a.N + n;
    h.rew = ldg<NT>(a.r + off); h.cost = ldg<NT>(a.c + off); h.vr = ldg<NT>(a.vr + off); h.vc = ldg<NT>(a.vc + off); h.d = ldg<NT>(a.d + off);
    h.lvr = a.lvr[n]; h.lvc = a.lvc[n]; h.ld = a.ld[n];
  } else {                                           // row t1 belongs to the next wave / workgroup, which streams it too: plain loads
    const size_t off = (size_t)t1 * a.N + n;
    h.vr = a.vr[off]; h.vc = a.vc[off]; h.d = a.d[off];
  }
}

// walk_chunk's head on values held in registers: row T - 1 bootstraps from the last values (`1.0 - last_dones(bool)` is float64 in the
// reference, buffers.py:530-531), every other chunk starts from row t1's values
template <bool WRITE, bool NT>
__device__ __forceinline__ void rs_head_apply(const GaeArgs& a, unsigned n, bool live, int t1, const HeadRegs& h, Carry& s) {
  if (t1 == a.T) {
    const double nnt = 1.0 - (h.ld ? 1.0 : 0.0);
    const float gvr = a.g_r * h.lvr;
    const float gvc = a.g_c * h.lvc;
    s.Ar = ((double)h.rew + (double)gvr * nnt) - (double)h.vr;
    s.Ac = ((double)h.cost + (double)gvc * nnt) - (double)h.vc;
    if (WRITE) {
      if (live) {
        const size_t off = (size_t)(a.T - 1) * a.N + n;
        const float fr = (float)s.Ar, fc = (float)s.Ac;
        stg<NT>(a.ar + off, fr); stg<NT>(a.ac + off, fc); stg<NT>(a.rr + off, fr + h.vr); stg<NT>(a.rc + off, fc + h.vc);
      }
    } else {
      s.Pr = 0.0;
      s.Pc = 0.0;
    }
  }
  s.vr_next = h.vr;
  s.vc_next = h.vc;
  s.d_next = h.d;
}

template <bool NT>
__device__ __forceinline__ void gae_dual_regsplit_body(const GaeArgs& a, int C, unsigned tag, double* maps, unsigned* flags, unsigned* status,
                                                       unsigned spin_limit, int fault, int tile_major = 0) {
  __shared__ double own[RS_W][4][64];
  __shared__ double ext[RS_CMAX - 1][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // CHUNK-major order: workgroups with consecutive blockIdx own ADJACENT column tiles of the same 128 rows — what runs at the same time reads
  // and writes whole rows (DRAM pages) instead of 256-byte pieces 4 N bytes apart, and the maps of the later chunks (lower blockIdx: a whole
  // sweep over the tiles earlier) have long been published when a workgroup asks for them.  (tile_major: the tile's chunks back to back, A/B)
  const int n_tiles = (int)gridDim.x / C;
  const int tile = tile_major ? (int)blockIdx.x / C : (int)blockIdx.x % n_tiles;
  const int c = C - 1 - (tile_major ? (int)blockIdx.x - tile * C : (int)blockIdx.x / n_tiles);       // time chunk of this workgroup: latest rows first
  const int col = tile * 64 + lane;
  const bool live = col < a.N;
  const unsigned n = live ? col : a.N - 1;
  const int wt0 = c * RS_TC, wt1 = wt0 + RS_TC < a.T ? wt0 + RS_TC : a.T;
  const int t0 = wt0 + wave * RS_R < wt1 ? wt0 + wave * RS_R : wt1;
  const int t1 = t0 + RS_R < wt1 ? t0 + RS_R : wt1;
  const bool any = t0 < t1;
  const int tb = t1 == a.T ? t1 - 2 : t1 - 1;          // first row of the register batch (row T - 1 is the head)
  Carry s;
  s.Ar = 0.0; s.Ac = 0.0; s.Pr = 1.0; s.Pc = 1.0;
  s.vr_next = 0.f; s.vc_next = 0.f; s.d_next = 0.f;
  HeadRegs h = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
  Batch<RS_R> b;
  if (any) {
    if (tb >= t0) load_batch<RS_R, NT>(a, n, tb, t0, b);
    rs_head_load<NT>(a, n, t1, h);
    rs_head_apply<false, NT>(a, n, live, t1, h, s);
    if (tb >= t0) run_batch<false, RS_R, NT>(a, n, live, tb, t0, b, s);
  }
  own[wave][0][lane] = s.Pr;
  own[wave][1][lane] = s.Ar;
  own[wave][2][lane] = s.Pc;
  own[wave][3][lane] = s.Ac;
  __syncthreads();
  typedef unsigned long long u64;
  if (wave == 0 && c > 0 && !(fault && c == C - 1)) {  // the map of rows [wt0, wt1) for the chunks before it (fault: test injection)
    double Pr = 1.0, Qr = 0.0, Pc = 1.0, Qc = 0.0;
    for (int w = RS_W - 1; w >= 0; --w) {
      Qr = own[w][1][lane] + own[w][0][lane] * Qr;  Pr = own[w][0][lane] * Pr;
      Qc = own[w][3][lane] + own[w][2][lane] * Qc;  Pc = own[w][2][lane] * Pc;
    }
    u64* m = reinterpret_cast<u64*>(maps + (size_t)(tile * C + c) * 256);
    __hip_atomic_store(m + lane, (u64)__double_as_longlong(Pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // write-through (sc1) stores
    __hip_atomic_store(m + 64 + lane, (u64)__double_as_longlong(Qr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(m + 128 + lane, (u64)__double_as_longlong(Pc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(m + 192 + lane, (u64)__double_as_longlong(Qc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every lane's stores acknowledged, then ONE flag
    if (lane == 0) __hip_atomic_store(&flags[tile * C + c], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int j = c + 1 + wave; j < C; j += RS_W) {
    unsigned polls = 0;
    while (__hip_atomic_load(&flags[tile * C + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      if (++polls >= spin_limit) { if (lane == 0 && status != nullptr) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    const u64* m = reinterpret_cast<const u64*>(maps + (size_t)(tile * C + j) * 256);
#pragma unroll
    for (int k = 0; k < 4; ++k)                          // agent-scope (L1-bypassing) loads: the bytes were stored write-through
      ext[j - c - 1][k][lane] = __longlong_as_double((long long)__hip_atomic_load(m + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  __syncthreads();
  double Ar = 0.0, Ac = 0.0;
  for (int j = C - 1; j > c; --j) {
    Ar = ext[j - c - 1][1][lane] + ext[j - c - 1][0][lane] * Ar;
    Ac = ext[j - c - 1][3][lane] + ext[j - c - 1][2][lane] * Ac;
  }
  for (int w = RS_W - 1; w > wave; --w) {
    Ar = own[w][1][lane] + own[w][0][lane] * Ar;
    Ac = own[w][3][lane] + own[w][2][lane] * Ac;
  }
  if (any) {
    s.Ar = Ar;
    s.Ac = Ac;
    rs_head_apply<true, NT>(a, n, live, t1, h, s);
    if (tb >= t0) run_batch<true, RS_R, NT>(a, n, live, tb, t0, b, s);
  }
}

template <bool NT>
__global__ void __launch_bounds__(64 * RS_W) gae_dual_regsplit_kernel(GaeArgs a, int C, unsigned tag, double* maps, unsigned* flags, unsigned* status,
                                                                      unsigned spin_limit, int fault, int tile_major) {
  gae_dual_regsplit_body<NT>(a, C, tag, maps, flags, status, spin_limit, fault, tile_major);
}


// several independent [T,N] rollouts of one shape in ONE launch: grid (tiles * C, n_runs), run = blockIdx.y
struct GaeRun {
  GaeArgs a;
  double* maps;
  unsigned* flags;
  unsigned* status;
};
__global__ void __launch_bounds__(64 * SPLIT_W) gae_dual_split_batch_kernel(const GaeRun* __restrict__ runs, int C, unsigned tag) {
  const GaeRun r = runs[blockIdx.y];
  gae_dual_split_body(r.a, C, tag, r.maps, r.flags, r.status);
}
template <bool NT>
__global__ void __launch_bounds__(64 * RS_W) gae_dual_regsplit_batch_kernel(const GaeRun* __restrict__ runs, int C, unsigned tag) {
  const GaeRun r = runs[blockIdx.y];
  gae_dual_regsplit_body<NT>(r.a, C, tag, r.maps, r.flags, r.status, GAE_SPIN_LIMIT, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// streaming shape with FOUR adjacent columns per lane: every load / store is a 16-byte dwordx4 (1 KB per wave-instruction and
// array instead of 256 B: a quarter of the memory instructions for the same bytes).  Same sequential per-column recurrence as the
// one-column kernel (bit-exact).  Needs N % 4 == 0 (rows stay 16-byte aligned).
// ---------------------------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
struct Batch4 {
  f4 r[U], c[U], vr[U], vc[U], d[U];
};

__device__ __forceinline__ f4 ldg4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)); }
__device__ __forceinline__ void stg4(float* p, f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); }

template <int U>
__device__ __forceinline__ void load_batch4(const GaeArgs& a, unsigned n, int t_top, Batch4<U>& b) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int t = t_top - u;
    t = t < 0 ? 0 : t;
    const size_t row = (size_t)t * a.N;
    b.r[u] = ldg4(a.r + row + n);
    b.c[u] = ldg4(a.c + row + n);
    b.vr[u] = ldg4(a.vr + row + n);
    b.vc[u] = ldg4(a.vc + row + n);
    b.d[u] = ldg4(a.d + row + n);
  }
}

struct Carry4 {
  double Ar[4], Ac[4];
  f4 vr_next, vc_next, d_next;
};

template <int U>
__device__ __forceinline__ void run_batch4(const GaeArgs& a, unsigned n, int t_top, const Batch4<U>& b, Carry4& s) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int t = t_top - u;
    if (t >= 0) {
      f4 fr, fc, rr, rc;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float nnt = 1.0f - s.d_next[k];
        const float gvr = (a.g_r * s.vr_next[k]) * nnt;      // ref: buffers.py:535-537 — float32 delta, float64 accumulate
        const float gvc = (a.g_c * s.vc_next[k]) * nnt;
        const float dr = (b.r[u][k] + gvr) - b.vr[u][k];
        const float dc = (b.c[u][k] + gvc) - b.vc[u][k];
        const float cr = a.gl_r * nnt;
        const float cc = a.gl_c * nnt;
        s.Ar[k] = (double)dr + (double)cr * s.Ar[k];
        s.Ac[k] = (double)dc + (double)cc * s.Ac[k];
        fr[k] = (float)s.Ar[k]; fc[k] = (float)s.Ac[k];
        rr[k] = fr[k] + b.vr[u][k]; rc[k] = fc[k] + b.vc[u][k];
      }
      const size_t row = (size_t)t * a.N;
      stg4(a.ar + row + n, fr);
      stg4(a.ac + row + n, fc);
      stg4(a.rr + row + n, rr);
      stg4(a.rc + row + n, rc);
      s.vr_next = b.vr[u];
      s.vc_next = b.vc[u];
      s.d_next = b.d[u];
    }
  }
}

template <int U, int G>
__global__ void __launch_bounds__(64 * G) gae_dual_x4_kernel(GaeArgs a) {
  // (an XCD-aware blockIdx -> column-range mapping was measured: no difference, every byte is touched once)
  const unsigned n = 4u * (blockIdx.x * 64 * G + threadIdx.x);
  if (n >= (unsigned)a.N) return;                      // N % 4 == 0: a lane's four columns are all inside or all outside
  Carry4 s;
  int t = a.T - 1;
  {  // t = T-1: bootstrap from the last values; `1.0 - last_dones(bool)` is float64 in the reference (buffers.py:530-531)
    const size_t off = (size_t)t * a.N + n;
    const f4 rew = ldg4(a.r + off), cost = ldg4(a.c + off), vr = ldg4(a.vr + off), vc = ldg4(a.vc + off), d = ldg4(a.d + off);
    f4 fr, fc, rr, rc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double nnt = 1.0 - (a.ld[n + k] ? 1.0 : 0.0);
      const float gvr = a.g_r * a.lvr[n + k];
      const float gvc = a.g_c * a.lvc[n + k];
      s.Ar[k] = ((double)rew[k] + (double)gvr * nnt) - (double)vr[k];
      s.Ac[k] = ((double)cost[k] + (double)gvc * nnt) - (double)vc[k];
      fr[k] = (float)s.Ar[k]; fc[k] = (float)s.Ac[k];
      rr[k] = fr[k] + vr[k]; rc[k] = fc[k] + vc[k];
    }
    stg4(a.ar + off, fr); stg4(a.ac + off, fc); stg4(a.rr + off, rr); stg4(a.rc + off, rc);
    s.vr_next = vr; s.vc_next = vc; s.d_next = d;
    --t;
  }
  if (t < 0) return;
  Batch4<U> b0, b1;
  load_batch4<U>(a, n, t, b0);
  while (true) {
    if (t - U >= 0) load_batch4<U>(a, n, t - U, b1);
    run_batch4<U>(a, n, t, b0, s);
    t -= U;
    if (t < 0) break;
    if (t - U >= 0) load_batch4<U>(a, n, t - U, b0);
    run_batch4<U>(a, n, t, b1, s);
    t -= U;
    if (t < 0) break;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Diagnostic (bench.py: `roofline.copy_gbs`): the memory traffic of gae_dual_x4_kernel<4, 1> WITHOUT its recurrence — the same grid
// (one wave per 256 columns walking the T rows from the last one down), the same five non-temporal 16-byte loads and four stores
// per lane and row, U rows in flight twice — so that a bench line can quote the GAE kernel against what the SAME box, process and
// stream deliver for the same bytes (HBM rate varies box to box more than the kernel does).  mode 1: a flat grid-stride 1:1 copy.
// ---------------------------------------------------------------------------------------------------------------------
template <int U>
__global__ void __launch_bounds__(64) stream_ref_x4_kernel(GaeArgs a) {
  const unsigned n = 4u * (blockIdx.x * 64 + threadIdx.x);
  if (n >= (unsigned)a.N) return;
  auto run = [&](int t_top, const Batch4<U>& b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t_top - u;
      if (t >= 0) {
        const size_t row = (size_t)t * a.N;
        stg4(a.ar + row + n, b.r[u] + b.d[u]);
        stg4(a.ac + row + n, b.c[u]);
        stg4(a.rr + row + n, b.vr[u]);
        stg4(a.rc + row + n, b.vc[u]);
      }
    }
  };
  int t = a.T - 1;
  Batch4<U> b0, b1;
  load_batch4<U>(a, n, t, b0);
  while (true) {
    if (t - U >= 0) load_batch4<U>(a, n, t - U, b1);
    run(t, b0);
    t -= U;
    if (t < 0) break;
    if (t - U >= 0) load_batch4<U>(a, n, t - U, b0);
    run(t, b1);
    t -= U;
    if (t < 0) break;
  }
}

__global__ void __launch_bounds__(256) stream_copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

// every split launch tags its workspace flags with a process-wide counter value, so nothing is cleared between launches
static unsigned next_split_tag() {
  static std::atomic<unsigned> launches{0};
  unsigned tag = ++launches;
  if (tag == 0) tag = ++launches;                  // 0 = the workspace's initial contents
  return tag;
}

extern "C" int icrl_gae_dual_ws(const float* rewards, const float* costs, const float* reward_values,
                                const float* cost_values, const float* dones, const float* last_v_r,
                                const float* last_v_c, const uint8_t* last_dones, float* adv_r, float* adv_c,
                                float* ret_r, float* ret_c, int T, int N, double reward_gamma, double reward_gae_lambda,
                                double cost_gamma, double cost_gae_lambda, int waves_per_tile, void* ws, long long ws_bytes,
                                void* stream) {
  if (T <= 0 || N <= 0) return fail("icrl_gae_dual: T = %d, N = %d", T, N);
  GaeArgs a{rewards, costs, reward_values, cost_values, dones, last_v_r, last_v_c, last_dones,
            adv_r, adv_c, ret_r, ret_c, T, N,
            (float)reward_gamma, (float)(reward_gamma * reward_gae_lambda),   // Python double product, then f32
            (float)cost_gamma, (float)(cost_gamma * cost_gae_lambda)};
  const int tiles = (N + 63) / 64;
  int W = waves_per_tile;
  if (W >= 100) W = 1;
  if (W != 1 && W != 4 && W != 16) {
    W = tiles >= 512 ? 1 : (tiles >= 64 ? 4 : 16);
    while (W > 1 && T / W < 2 * U_DEFAULT) W /= 4;
  }
  hipStream_t s = (hipStream_t)stream;
  // the last word of the workspace is the launch's STATUS word (0; 1: a bounded wait for another workgroup's map expired — the outputs are
  // then invalid; icrl_amd/buffers.py reads it next to the update's statistics)
  unsigned* status = ws != nullptr && ws_bytes >= 16 ? reinterpret_cast<unsigned*>((char*)ws + (ws_bytes & ~7ll) - 4) : nullptr;
  const long long ws_maps = status != nullptr ? (ws_bytes & ~7ll) - 8 : 0;
  // codes 300 + C / 501: the split scans with a chunk that never publishes and a short spin limit (fault injection for the tests)
  const int fault = (waves_per_tile >= 300 && waves_per_tile < 400) || waves_per_tile == 501;
  if (waves_per_tile >= 300 && waves_per_tile < 400) waves_per_tile -= 100;
  const unsigned spin_limit = fault ? 2048u : GAE_SPIN_LIMIT;
  // register-resident split scan (waves_per_tile 0 with enough workspace, or 500 / 501 to force it): T <= 2048, below the streaming shapes
  const int tile_major = waves_per_tile == 502;          // A/B: the tile's chunks on consecutive workgroups (500: chunk-major, the default)
  if (tile_major) waves_per_tile = 500;
  if (waves_per_tile == 0 || waves_per_tile == 500 || waves_per_tile == 501) {
    const int C = (T + RS_TC - 1) / RS_TC;
    const long long need = (long long)tiles * C * (256 * 8 + 4);
    const bool fits = C <= RS_CMAX && ws != nullptr && ws_maps >= need && (long long)tiles * C <= 0x7fffffffll;
    if (fits && (waves_per_tile != 0 || tiles < 1024)) {
      const unsigned tag = next_split_tag();
      double* maps = (double*)ws;
      unsigned* flags = (unsigned*)(maps + (size_t)tiles * C * 256);
      // non-temporal loads / stores once the arrays are beyond what the caches hold (> 128 tiles x 2048 rows x 36 B = 600 MB)
      if ((long long)tiles * T > 128ll * 2048) hipLaunchKernelGGL((gae_dual_regsplit_kernel<true>), dim3(tiles * C), dim3(64 * RS_W), 0, s, a, C, tag, maps, flags, status, spin_limit, fault, tile_major);
      else hipLaunchKernelGGL((gae_dual_regsplit_kernel<false>), dim3(tiles * C), dim3(64 * RS_W), 0, s, a, C, tag, maps, flags, status, spin_limit, fault, tile_major);
      return (int)hipGetLastError();
    }
    if (waves_per_tile != 0) return fail("icrl_gae_dual_ws: the register-resident split scan needs T <= %d and %lld B of workspace + 8 (T = %d, %lld given)", RS_TC * RS_CMAX, need, T, ws_bytes);
  }
  // up to 128 column tiles and a caller-owned workspace: the time axis is split over workgroups, two passes (waves_per_tile 0 when the
  // register-resident form does not apply — T > 2048 —, or 200 + C to force C workgroups per tile)
  if ((waves_per_tile == 0 || waves_per_tile >= 200) && waves_per_tile < 300 && ws != nullptr && tiles * 2 <= 256) {
    int C = waves_per_tile >= 200 ? waves_per_tile - 200 : T / (SPLIT_W * U_DEFAULT);
    if (C > SPLIT_CMAX) C = SPLIT_CMAX;
    // measured (tools/gae_small.py): 1 tile x 16, 4 tiles x 8 (T = 1024) / x 4 (T = 512), 8 tiles x 4 are the fastest splits:
    // about 32 workgroups for the small shapes; more only adds hops to the fold
    const int wg_cap = waves_per_tile >= 200 || tiles > 16 ? 256 : 32;
    while (C > 1 && tiles * C > wg_cap) --C;
    const long long need = (long long)tiles * C * (256 * 8 + 4);
    if (C >= 2 && ws_maps >= need) {
      const unsigned tag = next_split_tag();
      double* maps = (double*)ws;
      unsigned* flags = (unsigned*)(maps + (size_t)tiles * C * 256);
      hipLaunchKernelGGL(gae_dual_split_kernel, dim3(tiles * C), dim3(64 * SPLIT_W), 0, s, a, C, tag, maps, flags, status, spin_limit, fault);
      return (int)hipGetLastError();
    }
    if (waves_per_tile >= 200) return fail("icrl_gae_dual_ws: split %d needs 2..%d workgroups per tile and %lld B of workspace + 8 (%lld given)", waves_per_tile - 200, SPLIT_CMAX, need, ws_bytes);
  }
  // one wave per 64-env tile streams T rows; with >= 1024 tiles, 4 neighbouring tiles share a workgroup (1 KB contiguous per
  // row and array, the 4 waves start together) and 16 rows x 5 arrays are in flight per wave: +5..8 % of HBM rate measured.
  // waves_per_tile codes 101 / 105 / 106 force those shapes for tools/gae_variants.py.
  // >= 2048 tiles and N % 4 == 0: four columns per lane (dwordx4), one wave per workgroup, 4 rows x 5 arrays x 1 KB in flight twice:
  // 6.25-6.3 TB/s at 131 072 envs against 5.9-6.0 for shape 106 (tools/gae_variants.py); at 65 536 envs there is only one such
  // wave per CU and shape 106 is faster (6.0-6.1 against 5.4-5.7).
  int shape = waves_per_tile >= 100 ? waves_per_tile : (W == 1 ? (tiles >= 1024 ? (tiles >= 2048 && N % 4 == 0 ? 111 : 106) : 101) : W);
  if (shape == 101) hipLaunchKernelGGL((gae_dual_kernel<1, 8, true>), dim3(tiles), dim3(64), 0, s, a);
  else if (shape == 105) hipLaunchKernelGGL((gae_dual_kernel<1, 8, true, 4>), dim3((tiles + 3) / 4), dim3(256), 0, s, a);
  else if (shape == 106) hipLaunchKernelGGL((gae_dual_kernel<1, 16, true, 4>), dim3((tiles + 3) / 4), dim3(256), 0, s, a);
  else if (shape >= 107 && shape <= 112 && N % 4 == 0) {     // four columns per lane: 107 / 108 / 109 = U 2 / 4 / 8 with 4 waves, 110 / 111 / 112 with 1 wave
    const int lanes = (N / 4 + 63) / 64;                     // waves of 256 columns
    if (shape == 107) hipLaunchKernelGGL((gae_dual_x4_kernel<2, 4>), dim3((lanes + 3) / 4), dim3(256), 0, s, a);
    else if (shape == 108) hipLaunchKernelGGL((gae_dual_x4_kernel<4, 4>), dim3((lanes + 3) / 4), dim3(256), 0, s, a);
    else if (shape == 109) hipLaunchKernelGGL((gae_dual_x4_kernel<8, 4>), dim3((lanes + 3) / 4), dim3(256), 0, s, a);
    else if (shape == 110) hipLaunchKernelGGL((gae_dual_x4_kernel<2, 1>), dim3(lanes), dim3(64), 0, s, a);
    else if (shape == 111) hipLaunchKernelGGL((gae_dual_x4_kernel<4, 1>), dim3(lanes), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((gae_dual_x4_kernel<8, 1>), dim3(lanes), dim3(64), 0, s, a);
  }

  else if (shape == 4) hipLaunchKernelGGL((gae_dual_kernel<4, 8, false>), dim3(tiles), dim3(256), 0, s, a);
  else if (shape == 16) hipLaunchKernelGGL((gae_dual_kernel<16, 8, false>), dim3(tiles), dim3(1024), 0, s, a);
  else return fail("icrl_gae_dual_ex: waves_per_tile = %d (0 = automatic, 1, 4, 16 or a shape code 101 / 105 / 106, 107..112 with N %% 4 == 0)", waves_per_tile);
  return (int)hipGetLastError();
}

extern "C" size_t icrl_gae_dual_ws_bytes(int T, int N) {
  if (T <= 0 || N <= 0) return 0;
  const long long tiles = (N + 63) / 64, C = (T + RS_TC - 1) / RS_TC;
  long long need = ICRL_GAE_WS_BYTES;                                    // the two-pass split's maximum (256 maps)
  if (C <= RS_CMAX && tiles < 1024 && tiles * C * (256 * 8 + 4) > need - 8) need = tiles * C * (256 * 8 + 4) + 8;
  return (size_t)((need + 15) & ~7ll);                                   // + the status word, a multiple of 8
}

extern "C" int icrl_gae_dual_ex(const float* rewards, const float* costs, const float* reward_values,
                                const float* cost_values, const float* dones, const float* last_v_r,
                                const float* last_v_c, const uint8_t* last_dones, float* adv_r, float* adv_c,
                                float* ret_r, float* ret_c, int T, int N, double reward_gamma, double reward_gae_lambda,
                                double cost_gamma, double cost_gae_lambda, int waves_per_tile, void* stream) {
  return icrl_gae_dual_ws(rewards, costs, reward_values, cost_values, dones, last_v_r, last_v_c, last_dones, adv_r,
                          adv_c, ret_r, ret_c, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda,
                          waves_per_tile, nullptr, 0, stream);
}

extern "C" int icrl_gae_dual(const float* rewards, const float* costs, const float* reward_values,
                             const float* cost_values, const float* dones, const float* last_v_r,
                             const float* last_v_c, const uint8_t* last_dones, float* adv_r, float* adv_c,
                             float* ret_r, float* ret_c, int T, int N, double reward_gamma, double reward_gae_lambda,
                             double cost_gamma, double cost_gae_lambda, void* stream) {
  return icrl_gae_dual_ex(rewards, costs, reward_values, cost_values, dones, last_v_r, last_v_c, last_dones, adv_r,
                          adv_c, ret_r, ret_c, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0,
                          stream);
}

extern "C" int icrl_debug_stream_ref(const float* in0, const float* in1, const float* in2, const float* in3, const float* in4,
                                     float* out0, float* out1, float* out2, float* out3, int T, int N, int mode, void* stream) {
  if (T <= 0 || N <= 0 || N % 4 != 0) return fail("icrl_debug_stream_ref: T = %d, N = %d (N %% 4 == 0)", T, N);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) {
    GaeArgs a{in0, in1, in2, in3, in4, nullptr, nullptr, nullptr, out0, out1, out2, out3, T, N, 0.f, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL((stream_ref_x4_kernel<4>), dim3((N / 4 + 63) / 64), dim3(64), 0, s, a);
  } else {
    const float* src[4] = {in0, in1, in2, in3};
    float* dst[4] = {out0, out1, out2, out3};
    const size_t n4 = (size_t)T * N / 4;
    for (int k = 0; k < 4; ++k)
      hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 16), dim3(256), 0, s, reinterpret_cast<const f4*>(src[k]), reinterpret_cast<f4*>(dst[k]), n4);
  }
  return (int)hipGetLastError();
}

extern "C" int icrl_abi_version(void) { return 106; }

// icrl_gae_dual_ws for n_runs rollouts of one shape in ONE launch (the loop-size launches of several runs sharing a GPU): the
// two-level scan over workgroups, every run with its own workspace.  Shapes the split scan does not serve (> 128 column tiles, T too
// short to split, a run without workspace) are issued as n_runs single launches.
extern "C" int icrl_gae_dual_batch(int n_runs, const icrl_gae_job_t* jobs, int T, int N, double reward_gamma, double reward_gae_lambda,
                                   double cost_gamma, double cost_gae_lambda, void* args_ws, long long args_ws_bytes, void* stream) {
  static_assert(sizeof(GaeRun) <= ICRL_BATCH_ARGS_BYTES, "ICRL_BATCH_ARGS_BYTES");
  if (n_runs < 1 || n_runs > 65535) return fail("icrl_gae_dual_batch: n_runs = %d (1..65535)", n_runs);
  if (T <= 0 || N <= 0) return fail("icrl_gae_dual_batch: T = %d, N = %d", T, N);
  hipStream_t s = (hipStream_t)stream;
  const int tiles = (N + 63) / 64;
  // the single-launch heuristic of icrl_gae_dual_ws (same scan structure, same results): the register-resident split scan when it applies,
  // else the two-pass split
  const int Cr = (T + RS_TC - 1) / RS_TC;
  const bool regs = Cr <= RS_CMAX && tiles < 1024;
  int C = T / (SPLIT_W * U_DEFAULT);
  if (C > SPLIT_CMAX) C = SPLIT_CMAX;
  const int wg_cap = tiles > 16 ? 256 : 32;
  while (C > 1 && tiles * C > wg_cap) --C;
  if (regs) C = Cr;
  const long long need = (long long)tiles * C * (256 * 8 + 4);
  bool split = (regs || (tiles * 2 <= 256 && C >= 2)) && args_ws != nullptr && args_ws_bytes >= (long long)n_runs * ICRL_BATCH_ARGS_BYTES;
  for (int r = 0; r < n_runs && split; ++r) split = jobs[r].ws != nullptr && (jobs[r].ws_bytes & ~7ll) - 8 >= need;
  if (!split) {
    for (int r = 0; r < n_runs; ++r) {
      const icrl_gae_job_t& j = jobs[r];
      const int e = icrl_gae_dual_ws(j.rewards, j.costs, j.reward_values, j.cost_values, j.dones, j.last_v_r, j.last_v_c, j.last_dones,
                                     j.adv_r, j.adv_c, j.ret_r, j.ret_c, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0,
                                     j.ws, j.ws_bytes, stream);
      if (e) return e;
    }
    return 0;
  }
  GaeRun* d_runs = (GaeRun*)args_ws;
  for (int r = 0; r < n_runs; ++r) {
    const icrl_gae_job_t& j = jobs[r];
    GaeRun g;
    g.a = GaeArgs{j.rewards, j.costs, j.reward_values, j.cost_values, j.dones, j.last_v_r, j.last_v_c, j.last_dones,
                  j.adv_r, j.adv_c, j.ret_r, j.ret_c, T, N,
                  (float)reward_gamma, (float)(reward_gamma * reward_gae_lambda), (float)cost_gamma, (float)(cost_gamma * cost_gae_lambda)};
    g.maps = (double*)j.ws;
    g.flags = (unsigned*)(g.maps + (size_t)tiles * C * 256);
    g.status = reinterpret_cast<unsigned*>((char*)j.ws + (j.ws_bytes & ~7ll) - 4);
    const int e = icrl::put_args(g, d_runs + r, s);
    if (e) return e;
  }
  if (!regs) hipLaunchKernelGGL(gae_dual_split_batch_kernel, dim3(tiles * C, n_runs), dim3(64 * SPLIT_W), 0, s, d_runs, C, next_split_tag());
  else if ((long long)tiles * T > 128ll * 2048) hipLaunchKernelGGL((gae_dual_regsplit_batch_kernel<true>), dim3(tiles * C, n_runs), dim3(64 * RS_W), 0, s, d_runs, C, next_split_tag());
  else hipLaunchKernelGGL((gae_dual_regsplit_batch_kernel<false>), dim3(tiles * C, n_runs), dim3(64 * RS_W), 0, s, d_runs, C, next_split_tag());
  return (int)hipGetLastError();
}

namespace icrl {
int icrl_gae_dual_batch_impl(int n_runs, const icrl_rollout_job_t* jobs, double reward_gamma, double reward_gae_lambda, double cost_gamma,
                             double cost_gae_lambda, void* args_ws, void* stream) {
  icrl_gae_job_t stack_jobs[64];
  icrl_gae_job_t* gj = n_runs <= 64 ? stack_jobs : new icrl_gae_job_t[n_runs];
  for (int r = 0; r < n_runs; ++r) {
    const icrl_buffer_t* b = jobs[r].buf;
    const icrl_agent_t* ag = jobs[r].ag;
    gj[r] = icrl_gae_job_t{b->rewards, b->costs, b->reward_values, b->cost_values, b->dones, ag->last_v_r, ag->last_v_c, ag->last_dones,
                           b->reward_advantages, b->cost_advantages, b->reward_returns, b->cost_returns, b->gae_ws, b->gae_ws_bytes};
  }
  const int e = icrl_gae_dual_batch(n_runs, gj, jobs[0].buf->T, jobs[0].buf->N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda,
                                    args_ws, (long long)n_runs * ICRL_BATCH_ARGS_BYTES, stream);
  if (gj != stack_jobs) delete[] gj;
  return e;
}
}  // namespace icrl
