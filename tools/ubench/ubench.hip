// Micro-benchmarks of the building blocks of the PPO update kernel on gfx950 (one wave per SIMD unless stated):
// cycles (s_memtime, 100 MHz-independent shader clock) per operation.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ float fast_tanh(float x) {
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  const float e = __expf(2.f * xc);
  return (e - 1.f) * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float fast_tanh2(float x) {   // 1 - 2 / (exp2(c x) + 1)
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

// Is a chain of fp32 16x16x4 MFMAs bit-identical to the sequential chain acc = fmaf(a[k], b[k], acc), k ascending?  A is 16 x K,
// B is K x 16 (K = 64); mismatching output elements are counted.  scale mixes magnitudes so that rounding differs between orders.
__global__ void k_mfma_vs_fma(const float* A, const float* B, int K, int* mismatches, float* worst) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  f32x4 acc = f32x4{0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) acc = MFMA(A[r * K + k0 + q], B[(k0 + q) * 16 + r], acc);
  int bad = 0; float wdev = 0.f;
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * q + i, col = r;
    float ref = 0.f;
    for (int k = 0; k < K; ++k) ref = fmaf(A[row * K + k], B[k * 16 + col], ref);
    if (__float_as_uint(ref) != __float_as_uint(acc[i])) { ++bad; wdev = fmaxf(wdev, fabsf(ref - acc[i])); }
  }
  atomicAdd(mismatches, bad);
  atomicMax(reinterpret_cast<unsigned*>(worst), __float_as_uint(wdev));
}

template <int CHAINS>
__global__ void k_mfma(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = MFMA(a, b, acc[c]);
  }
  unsigned long long t1 = now();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WHICH>
__global__ void k_tanh(float* out, unsigned long long* cyc, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = WHICH == 0 ? fast_tanh(v[u] + 0.1f) : fast_tanh2(v[u] + 0.1f);
  unsigned long long t1 = now();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// MFMA interleaved with independent tanh work: does the VALU work hide under the matrix pipe?
__global__ void k_mix(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = MFMA(a, b, acc[c]);
      v[u] = fast_tanh2(v[u] + 0.1f);
    }
  }
  unsigned long long t1 = now();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same mix with the interleaving forced: 1 MFMA, then up to 2 VALU, ... (sched_group_barrier masks: 0x8 MFMA, 0x2 VALU)
__global__ void k_mix_sched(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = MFMA(a, b, acc[c]);
      v[u] = fast_tanh2(v[u] + 0.1f);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
      }
    }
  }
  unsigned long long t1 = now();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 4 MFMA + 16 independent plain VALU ops (4 per MFMA shadow)
__global__ void k_mix_valu(float* out, unsigned long long* cyc, int iters, int mode) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = MFMA(a, b, acc[c]);
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], 1.0001f, 0.5f);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
      }
    }
  }
  unsigned long long t1 = now();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// two waves per SIMD in different phases: waves 0-3 only MFMA, waves 4-7 only VALU (does the VALU work of ANOTHER wave hide?)
__global__ void k_split(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  const bool matrix = threadIdx.x < 256;
  unsigned long long t0 = now();
  if (matrix) {
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MFMA(a, b, acc[c]);
  } else {
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 32; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], 1.0001f, 0.5f);
  }
  unsigned long long t1 = now();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 255) == 0) cyc[threadIdx.x >> 8] = t1 - t0;
}

// float64: dependent add chain, division, sqrt (the normaliser's statistics are a serial float64 recurrence)
__global__ void k_f64(double* out, unsigned long long* cyc, int iters, int which) {
  double v = 1.0 + threadIdx.x * 1e-3, u = 3.0 + threadIdx.x;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (which == 0) v = v + u;
      else if (which == 1) v = v / u + 1.5;
      else if (which == 2) v = sqrt(v + u);
      else v = fma(v, u, 0.25);
    }
  }
  unsigned long long t1 = now();
  out[threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_f64_lds(double* out, unsigned long long* cyc, int iters) {
  __shared__ double sm[64 * 18];
  for (int i = threadIdx.x; i < 64 * 18; i += blockDim.x) sm[i] = i * 0.5;
  __syncthreads();
  double acc = 0.0;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
    double sum = 0.0;
    if (threadIdx.x < 18) {
#pragma unroll 8
      for (int r = 0; r < 64; ++r) sum += sm[r * 18 + threadIdx.x];
    }
    acc += sum;
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = now();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// LDS: batch of NB ds_read_b128 with the operand pattern of the kernel (row stride S), then use
template <int NB, int S>
__global__ void k_lds128(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 64 * S; i += blockDim.x) sm[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  float s = 0;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
    f32x4 v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = *reinterpret_cast<const f32x4*>(sm + ((u / 4) * 16 + r) * S + 16 * (u % 4) + 4 * q);
#pragma unroll
    for (int u = 0; u < NB; ++u) s += v[u][0] + v[u][3];
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = now();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// transposed b32 reads (dH1 A operand): NB reads at (16 js + 4 q + e) * S + 16 t + r
template <int NB, int S>
__global__ void k_lds32(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 64 * S; i += blockDim.x) sm[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  float s = 0;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
    float v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = sm[(16 * ((u / 4) % 4) + 4 * q + (u % 4)) * S + 16 * (u / 16) + r];
#pragma unroll
    for (int u = 0; u < NB; ++u) s += v[u];
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = now();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// v_fmac_f32: CH independent accumulator chains of 64 terms each, weights in registers; MODE 0 = activations in registers,
// 1 = activations fetched by broadcast ds_read_b128 (every lane the same address, the rollout kernels' pattern), prefetched one
// group ahead, 2 = the same without prefetch (load, wait, use)
template <int CH, int MODE>
__global__ void k_fmac(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 1e-3f * (i & 31);
  __syncthreads();
  float w[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) w[k] = 1e-3f * (threadIdx.x + k);
  float acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = c;
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 64; ++k)
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c]) : "v"(w[k]), "v"(w[(k + c + 1) & 63]));
    } else {
      f32x4 cur[CH], nxt[CH];
      const float* base = sm + (i & 7) * 64 * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) cur[c] = *reinterpret_cast<const f32x4*>(base + c * 64);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (MODE == 1 && j < 15) {
#pragma unroll
          for (int c = 0; c < CH; ++c) nxt[c] = *reinterpret_cast<const f32x4*>(base + c * 64 + 4 * (j + 1));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CH; ++c) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c]) : "v"(cur[c][e]), "v"(w[4 * j + e]));
        if (MODE == 2 && j < 15) {
#pragma unroll
          for (int c = 0; c < CH; ++c) nxt[c] = *reinterpret_cast<const f32x4*>(base + c * 64 + 4 * (j + 1));
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
      }
    }
  }
  unsigned long long t1 = now();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_barrier(float* out, unsigned long long* cyc, int iters) {
  unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned long long t1 = now();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[threadIdx.x] = 0;
}

// agent-scope granule round trip between two workgroups (ping-pong)
__global__ void k_pingpong(unsigned long long* flag, unsigned long long* cyc, int iters, int other = 1) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x != 0 && (int)blockIdx.x != other) return;
  const int me = blockIdx.x == 0 ? 0 : 1;
  unsigned long long t0 = now();
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      __hip_atomic_store(flag, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flag + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) {}
    } else {
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) {}
      __hip_atomic_store(flag + 16, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  unsigned long long t1 = now();
  cyc[me] = t1 - t0;
}


// bulk exchange between two workgroups (256 threads each): every thread publishes K values and reads the partner's K.
// MODE 0: 8-byte granules {step tag | float}, one agent-scope relaxed atomic store / load per value.
// MODE 1: 16-byte self-validating records {tag, a, b, tag}: two values per store / load (sc0 sc1 dwordx4), both tags checked.
template <int MODE, int K>
__global__ void k_bulk(unsigned long long* buf, unsigned long long* cyc, int iters, float* sink) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  float acc = 0.f;
  unsigned long long t0 = 0;
  for (int it = 1; it <= iters; ++it) {
    if (it == 2) t0 = __builtin_amdgcn_s_memtime();
    const unsigned tag = (unsigned)it;
    const size_t par = (size_t)(it & 1) * 2;
    if (MODE == 0) {
      unsigned long long* mine = buf + (par + wg) * (size_t)(K * 256) + tid;
      const unsigned long long* theirs = buf + (par + (1 - wg)) * (size_t)(K * 256) + tid;
#pragma unroll
      for (int k = 0; k < K; ++k)
        __hip_atomic_store(mine + (size_t)k * 256, ((unsigned long long)tag << 32) | (unsigned)__float_as_uint((float)(k + it)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // all K loads in flight together, one wait, then the tags; repeated until every granule carries this exchange's tag
      unsigned long long v[K];
      bool ok = false;
      for (int rounds = 0; rounds < (1 << 20) && !ok; ++rounds) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __hip_atomic_load(theirs + (size_t)k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && (unsigned)(v[k] >> 32) == tag;
        ok = __all(ok);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) acc += __uint_as_float((unsigned)v[k]);
    } else {
      u4* mine = reinterpret_cast<u4*>(buf) + (par + wg) * (size_t)(K / 2 * 256) + tid;
      const u4* theirs = reinterpret_cast<const u4*>(buf) + (par + (1 - wg)) * (size_t)(K / 2 * 256) + tid;
#pragma unroll
      for (int k = 0; k < K / 2; ++k) {
        u4 v = {tag, __float_as_uint((float)(2 * k + it)), __float_as_uint((float)(2 * k + 1 + it)), tag};
        u4* p = mine + (size_t)k * 256;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
      }
      u4 v[K / 2];
      bool ok = false;
      for (int rounds = 0; rounds < (1 << 20) && !ok; ++rounds) {
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
          const u4* p = theirs + (size_t)k * 256;
          asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = true;
#pragma unroll
        for (int k = 0; k < K / 2; ++k) { asm volatile("" : "+v"(v[k])); ok = ok && v[k][0] == tag && v[k][3] == tag; }
        ok = __all(ok);
      }
#pragma unroll
      for (int k = 0; k < K / 2; ++k) acc += __uint_as_float(v[k][1]) + __uint_as_float(v[k][2]);
    }
    __syncthreads();
  }
  if (tid == 0 && wg == 0) cyc[0] = __builtin_amdgcn_s_memtime() - t0;
  if (acc == 12345.f) sink[0] = acc;
}

// per-XCD speed: one wave per workgroup runs the same dependent MFMA chain; wall time from the constant-rate counter (s_memrealtime,
// 100 MHz) next to the shader-clock count (s_memtime) and the XCD the workgroup landed on
__global__ void k_xcd_speed(unsigned long long* out, int iters) {
  f32x4 acc = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = MFMA(a, b, acc);
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    out[4 * blockIdx.x + 0] = xcc; out[4 * blockIdx.x + 1] = r1 - r0; out[4 * blockIdx.x + 2] = c1 - c0;
    out[4 * blockIdx.x + 3] = (unsigned long long)(acc[0] == 12345.f);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  float* out; unsigned long long* cyc; unsigned long long* flag;
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 4096)); CK(hipMalloc(&flag, 4096));
  CK(hipMemset(flag, 0, 4096));
  std::vector<unsigned long long> h(16);
  const int it = 200;
  auto rd = [&](const char* name, double per) { hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost); printf("%-46s %8.1f cycles\n", name, (double)h[0] / per); return 0; };
  hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("mfma 16x16x4 f32, 1 dependent chain, 1 wave", it * 16.0);
  hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("mfma, 2 chains, per mfma", it * 32.0);
  hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("mfma, 4 chains, per mfma", it * 64.0);
  hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(256), 0, 0, out, cyc, it); rd("mfma, 4 chains, 4 waves (1/SIMD), per mfma", it * 64.0);
  hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(512), 0, 0, out, cyc, it); rd("mfma, 4 chains, 8 waves (2/SIMD), per mfma/wave", it * 64.0);
  hipLaunchKernelGGL(k_tanh<0>, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("fast_tanh (clamp, exp, rcp) per value", it * 16.0);
  hipLaunchKernelGGL(k_tanh<1>, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("fast_tanh2 (exp2, rcp, fma) per value", it * 16.0);
  hipLaunchKernelGGL(k_mix, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("4 mfma + 1 tanh2 interleaved, per group", it * 16.0);
  hipLaunchKernelGGL(k_mix_sched, dim3(1), dim3(64), 0, 0, out, cyc, it); rd("4 mfma + 1 tanh2, forced interleave, per group", it * 16.0);
  hipLaunchKernelGGL(k_mix_valu, dim3(1), dim3(64), 0, 0, out, cyc, it, 0); rd("4 mfma + 16 fma forced interleave (4 per mfma)", it * 4.0);
  { double* dout; hipMalloc(&dout, 4096);
    hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, 0, dout, cyc, it, 0); rd("f64 dependent add, per op", it * 16.0);
    hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, 0, dout, cyc, it, 3); rd("f64 dependent fma, per op", it * 16.0);
    hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, 0, dout, cyc, it, 1); rd("f64 division (+add), per op", it * 16.0);
    hipLaunchKernelGGL(k_f64, dim3(1), dim3(64), 0, 0, dout, cyc, it, 2); rd("f64 sqrt (+add), per op", it * 16.0);
    hipLaunchKernelGGL(k_f64_lds, dim3(1), dim3(64), 0, 0, dout, cyc, it); rd("64-row sequential f64 column sum from LDS", it); }
  hipLaunchKernelGGL(k_split, dim3(1), dim3(512), 0, 0, out, cyc, it); hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
  printf("split phases, 2 waves/SIMD: matrix waves %.0f cycles / 64 mfma (alone: 2048), vector waves %.0f cycles / 512 fma (alone: ~2048)\n", (double)h[0] / it, (double)h[1] / it);
  hipLaunchKernelGGL((k_lds128<16, 72>), dim3(1), dim3(64), 72 * 64 * 4, 0, out, cyc, it); rd("16 x ds_read_b128 stride 72, 1 wave, per batch", it);
  hipLaunchKernelGGL((k_lds128<16, 72>), dim3(1), dim3(256), 72 * 64 * 4, 0, out, cyc, it); rd("16 x ds_read_b128 stride 72, 4 waves, per batch", it);
  hipLaunchKernelGGL((k_lds128<16, 68>), dim3(1), dim3(256), 72 * 64 * 4, 0, out, cyc, it); rd("16 x ds_read_b128 stride 68, 4 waves, per batch", it);
  hipLaunchKernelGGL((k_lds128<4, 72>), dim3(1), dim3(256), 72 * 64 * 4, 0, out, cyc, it); rd("4 x ds_read_b128 stride 72, 4 waves, per batch", it);
  hipLaunchKernelGGL((k_lds32<64, 72>), dim3(1), dim3(256), 72 * 64 * 4, 0, out, cyc, it); rd("64 x ds_read_b32 transposed stride 72, 4 waves", it);
  hipLaunchKernelGGL((k_lds32<64, 68>), dim3(1), dim3(256), 72 * 64 * 4, 0, out, cyc, it); rd("64 x ds_read_b32 transposed stride 68, 4 waves", it);
  hipLaunchKernelGGL((k_lds32<16, 72>), dim3(1), dim3(256), 72 * 64 * 4, 0, out, cyc, it); rd("16 x ds_read_b32 transposed stride 72, 4 waves", it);
  hipLaunchKernelGGL((k_fmac<1, 0>), dim3(1), dim3(64), 16384, 0, out, cyc, it); rd("v_fmac, 1 dependent chain, registers, per fmac", it * 64.0);
  hipLaunchKernelGGL((k_fmac<2, 0>), dim3(1), dim3(64), 16384, 0, out, cyc, it); rd("v_fmac, 2 chains, registers, per fmac", it * 128.0);
  hipLaunchKernelGGL((k_fmac<4, 0>), dim3(1), dim3(64), 16384, 0, out, cyc, it); rd("v_fmac, 4 chains, registers, per fmac", it * 256.0);
  hipLaunchKernelGGL((k_fmac<4, 0>), dim3(1), dim3(256), 16384, 0, out, cyc, it); rd("v_fmac, 4 chains, registers, 4 waves, per fmac/wave", it * 256.0);
  hipLaunchKernelGGL((k_fmac<4, 1>), dim3(1), dim3(64), 16384, 0, out, cyc, it); rd("v_fmac, 4 chains, broadcast b128 prefetched, 1 wave", it * 256.0);
  hipLaunchKernelGGL((k_fmac<4, 1>), dim3(1), dim3(256), 16384, 0, out, cyc, it); rd("v_fmac, 4 chains, broadcast b128 prefetched, 4 waves", it * 256.0);
  hipLaunchKernelGGL((k_fmac<4, 2>), dim3(1), dim3(256), 16384, 0, out, cyc, it); rd("v_fmac, 4 chains, broadcast b128 not prefetched, 4 waves", it * 256.0);
  hipLaunchKernelGGL((k_fmac<2, 1>), dim3(1), dim3(256), 16384, 0, out, cyc, it); rd("v_fmac, 2 chains, broadcast b128 prefetched, 4 waves", it * 128.0);
  hipLaunchKernelGGL((k_fmac<1, 1>), dim3(1), dim3(256), 16384, 0, out, cyc, it); rd("v_fmac, 1 chain, broadcast b128 prefetched, 4 waves", it * 64.0);
  for (int blocks : {1, 8, 64, 256, 1024}) {   // does the rate hold when the whole chip runs the same chains (clock under load)?
    char nm[96]; snprintf(nm, sizeof nm, "v_fmac, 4 chains, broadcast b128, 4 waves, %d workgroups", blocks);
    hipLaunchKernelGGL((k_fmac<4, 1>), dim3(blocks), dim3(256), 16384, 0, out, cyc, 20 * it); rd(nm, 20 * it * 256.0);
  }
  { // bit-identity of MFMA accumulation with the fmaf chain
    const int K = 64; std::vector<float> hA(16 * K), hB(K * 16);
    unsigned st = 12345u; auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.f - 1.f; };
    float *dA, *dB, *dw; int* dm; CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dm, 4)); CK(hipMalloc(&dw, 4));
    int total = 0, trials = 200; float worst = 0.f;
    for (int tr = 0; tr < trials; ++tr) {
      for (auto& v : hA) { v = rnd(); if (tr & 1) v *= (rnd() > 0.5f ? 1e3f : 1e-3f); }
      for (auto& v : hB) v = rnd();
      hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
      hipMemset(dm, 0, 4); hipMemset(dw, 0, 4);
      hipLaunchKernelGGL(k_mfma_vs_fma, dim3(1), dim3(64), 0, 0, dA, dB, K, dm, dw);
      int m; float w; hipMemcpy(&m, dm, 4, hipMemcpyDeviceToHost); hipMemcpy(&w, dw, 4, hipMemcpyDeviceToHost);
      total += m; worst = w > worst ? w : worst;
    }
    printf("mfma 16x16x4 f32 chain vs sequential fmaf chain (K = 64): %d of %d outputs differ bitwise (worst |d| %.3e)\n", total, trials * 256, worst);
  }
  { // does the clock hold when every SIMD of every CU issues fp32 MFMAs for tens of milliseconds? (wall time of the same chain)
    unsigned long long* xo2; CK(hipMalloc(&xo2, 1024 * 4 * 4 * 8));
    for (int blocks : {3, 96, 192, 256}) {
      hipLaunchKernelGGL(k_xcd_speed, dim3(blocks), dim3(256), 0, 0, xo2, 200000);
      std::vector<unsigned long long> hx(4 * blocks); hipDeviceSynchronize(); hipMemcpy(hx.data(), xo2, hx.size() * 8, hipMemcpyDeviceToHost);
      double mx = 0, mn = 1e30; for (int b = 0; b < blocks; ++b) { const double us = hx[4 * b + 1] / 100.0; mx = us > mx ? us : mx; mn = us < mn ? us : mn; }
      printf("3.2 M dependent MFMAs per wave, %3d workgroups x 4 waves (one per SIMD): wall %.0f .. %.0f us per workgroup (102.4 M shader cycles = %.0f us at 2.4 GHz)\n", blocks, mn, mx, 102.4e6 / 2400.0);
    }
  }
  { // do the XCDs run at the same speed?
    unsigned long long* xo; CK(hipMalloc(&xo, 64 * 4 * 8));
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_xcd_speed, dim3(64), dim3(64), 0, 0, xo, 20000);
      std::vector<unsigned long long> hx(64 * 4); hipDeviceSynchronize(); hipMemcpy(hx.data(), xo, hx.size() * 8, hipMemcpyDeviceToHost);
      double best[8], worst[8]; int cnt[8];
      for (int x = 0; x < 8; ++x) { best[x] = 1e30; worst[x] = 0; cnt[x] = 0; }
      for (int b = 0; b < 64; ++b) { const int x = (int)hx[4 * b] & 7; const double us = hx[4 * b + 1] / 100.0; if (us < best[x]) best[x] = us; if (us > worst[x]) worst[x] = us; ++cnt[x]; }
      printf("320 k dependent MFMAs (10.24 M shader cycles), 64 workgroups, wall us by XCD (min..max of its workgroups):");
      for (int x = 0; x < 8; ++x) printf("  xcd%d[%d] %.0f..%.0f", x, cnt[x], best[x], worst[x]);
      printf("\n");
    }
  }
  hipLaunchKernelGGL(k_barrier, dim3(1), dim3(256), 0, 0, out, cyc, it); rd("s_barrier, 4 waves", it);
  hipLaunchKernelGGL(k_barrier, dim3(1), dim3(512), 0, 0, out, cyc, it); rd("s_barrier, 8 waves", it);
  hipLaunchKernelGGL(k_pingpong, dim3(2), dim3(64), 0, 0, flag, cyc, it, 1); rd("granule round trip, workgroups 0 and 1 (different XCDs)", it);
  CK(hipMemset(flag, 0, 4096));
  hipLaunchKernelGGL(k_pingpong, dim3(9), dim3(64), 0, 0, flag, cyc, it, 8); rd("granule round trip, workgroups 0 and 8 (same XCD)", it);
  CK(hipMemset(flag, 0, 4096));
  hipLaunchKernelGGL(k_pingpong, dim3(17), dim3(64), 0, 0, flag, cyc, it, 16); rd("granule round trip, workgroups 0 and 16 (same XCD)", it);
  // shader clock vs wall clock
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(64), 0, 0, out, cyc, 20000); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
  const unsigned long long clock_ticks = h[0];
  {
    unsigned long long* bulk;
    CK(hipMalloc(&bulk, 4 * 32 * 256 * 8 * 2));
    const int bi = 2000;
    auto run = [&](auto kern, const char* name, int K) {
      hipMemset(bulk, 0, 4 * 32 * 256 * 8 * 2);
      hipLaunchKernelGGL(kern, dim3(2), dim3(256), 0, 0, bulk, cyc, bi, out);
      hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
      printf("%-60s %8.1f cycles per exchange (%d values per thread each way)\n", name, (double)h[0] / (bi - 1), K);
    };
    run(k_bulk<0, 8>, "bulk exchange, 8-byte granules, 8 per thread", 8);
    run(k_bulk<0, 16>, "bulk exchange, 8-byte granules, 16 per thread", 16);
    run(k_bulk<0, 32>, "bulk exchange, 8-byte granules, 32 per thread", 32);
    run(k_bulk<1, 8>, "bulk exchange, 16-byte records (2 values), 8 per thread", 8);
    run(k_bulk<1, 16>, "bulk exchange, 16-byte records (2 values), 16 per thread", 16);
    run(k_bulk<1, 32>, "bulk exchange, 16-byte records (2 values), 32 per thread", 32);
  }
  printf("s_memtime ticks per us: %.1f  (ticks %llu over %.3f ms)\n", (double)clock_ticks / (ms * 1e3), clock_ticks, ms);
  return 0;
}
