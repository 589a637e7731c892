// Shared device helpers for libicrl_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/icrl_hip.h"

namespace icrl {

constexpr int WAVE = 64;
constexpr int MAX_OBS = 128;   // obs_dim limit of the rollout kernels (HC 18, Ant 113)
constexpr int MAX_ACT = 16;    // act_dim limit (HC 6, Ant 8)
constexpr int MAX_H = 64;      // hidden width limit of policy / cost nets
constexpr int MAX_CN_IN = 160; // cost-net input limit (Ant: 121)

// gae.hip: the batched dual-GAE launch behind icrl_rollout_collect_batch (jobs' buffers / agent state name the arrays)
int icrl_gae_dual_batch_impl(int n_runs, const icrl_rollout_job_t* jobs, double reward_gamma, double reward_gae_lambda, double cost_gamma,
                             double cost_gae_lambda, void* args_ws, void* stream);

// generic.hip: the generic-shape path (hidden widths above MAX_H up to 256, any MlpExtractor architecture given by icrl_policy_t.arch,
// any batch size) behind icrl_policy_forward / icrl_policy_evaluate / icrl_ppo_lag_train
int launch_policy_generic(const icrl_policy_t* p, const double* obs, const float* noise, int N, int deterministic, const float* alow,
                          const float* ahigh, float* actions, float* act_clipped, float* v_r, float* v_c, float* log_prob,
                          const float* given, float* entropy, hipStream_t s);
int launch_train_generic(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, const icrl_buffer_t* buf,
                         const int* perm_off, const float* nu, const icrl_ppo_hyper_t* hp, float* stats, void* scratch, void* sync_ws, hipStream_t s);
int generic_row_floats(const icrl_policy_t* pol);
int launch_generic_transpose(const icrl_policy_t* p, hipStream_t s);       // icrl_policy_prepare of a generic-path policy
int policy_generic_check(const icrl_policy_t* p, const char* who);      // 0, or the fail() code of an architecture the path refuses
inline bool policy_is_wide(const icrl_policy_t* p) { return p->arch != nullptr || p->h1 > MAX_H || p->h2 > MAX_H; }
// cn_train.hip: cost / discriminator forward of a constraint net with a hidden layer above MAX_H units or more than two of them (64 rows per workgroup)
int launch_cn_cost_rows(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* out, int mode, hipStream_t s);
inline bool costnet_is_wide(const icrl_costnet_t* cn) { return cn->n_hidden > 2 || cn->n_hidden == 0 || cn->h1 > MAX_H || (cn->n_hidden == 2 && cn->h2 > MAX_H); }

// argument rejection: formats the reason into the calling thread's icrl_last_error() text, returns hipErrorInvalidValue
int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// ---------------------------------------------------------------------------------------------------------------
// argument blocks of a BATCHED launch (several independent runs in one grid, run = blockIdx.y): the kernels read block
// blockIdx.y of an array in device memory.  Each block gets there through one tiny launch that carries it by value — stream-ordered
// like everything else, no host staging buffer whose lifetime would have to outlive the call, no hidden synchronisation.
// ---------------------------------------------------------------------------------------------------------------
// A pointer read from an argument block in MEMORY (batched launches) is a generic pointer to the compiler: every access through it
// becomes flat_load / flat_store, which count on the LDS counter too — an LDS wait then also waits for the global loads in flight
// (measured: the batched update step 9.5 us against 8.9 us for the single-run kernel whose by-value pointers are known to be global).
// Routing the pointer through the global address space once, where it is read, gives every access behind it the global_* form.
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  // UNIFORM pointers only (everything in an argument block is): the two halves go through readfirstlane — that also keeps the
  // generic -> global -> generic cast pair from being folded away — and the pointer lives in scalar registers from here on
  typedef T __attribute__((address_space(1)))* G;
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(G)(((unsigned long long)hi << 32) | (unsigned long long)lo);
#else
  return p;
#endif
}

// the same for every pointer of a C-ABI struct that a kernel copied out of an argument block in memory
__device__ __forceinline__ void globalize(icrl_env_t& e) {
  e.B = as_global(e.B); e.s = as_global(e.s); e.t_ep = as_global(e.t_ep); e.step_count = as_global(e.step_count); e.key = as_global(e.key);
}
__device__ __forceinline__ void globalize(icrl_norm_t& n) {
  n.obs_mean = as_global(n.obs_mean); n.obs_var = as_global(n.obs_var); n.obs_count = as_global(n.obs_count);
  n.ret_stats = as_global(n.ret_stats); n.cost_stats = as_global(n.cost_stats); n.ret = as_global(n.ret); n.cost_ret = as_global(n.cost_ret);
}
__device__ __forceinline__ void globalize(icrl_buffer_t& b) {
  b.observations = as_global(b.observations); b.new_observations = as_global(b.new_observations);
  b.orig_observations = as_global(b.orig_observations); b.new_orig_observations = as_global(b.new_orig_observations);
  b.actions = as_global(b.actions); b.dones = as_global(b.dones); b.log_probs = as_global(b.log_probs); b.rewards = as_global(b.rewards);
  b.reward_values = as_global(b.reward_values); b.costs = as_global(b.costs); b.orig_costs = as_global(b.orig_costs);
  b.cost_values = as_global(b.cost_values); b.reward_advantages = as_global(b.reward_advantages); b.reward_returns = as_global(b.reward_returns);
  b.cost_advantages = as_global(b.cost_advantages); b.cost_returns = as_global(b.cost_returns);
}
__device__ __forceinline__ void globalize(icrl_agent_t& g) {
  g.last_obs = as_global(g.last_obs); g.last_dones = as_global(g.last_dones); g.raw_rew = as_global(g.raw_rew); g.raw_cost = as_global(g.raw_cost);
  g.dones = as_global(g.dones); g.last_v_r = as_global(g.last_v_r); g.last_v_c = as_global(g.last_v_c); g.act_clipped = as_global(g.act_clipped);
  g.status = as_global(g.status);
}
__device__ __forceinline__ void globalize(icrl_costnet_t& c) {
  c.select_dim = as_global(c.select_dim); c.action_low = as_global(c.action_low); c.action_high = as_global(c.action_high);
  c.obs_mean = as_global(c.obs_mean); c.obs_var = as_global(c.obs_var); c.params = as_global(c.params); c.params_t = as_global(c.params_t);
}

// SINGLE-RUN persistent launches (their workgroups spin on each other's records / granules): a cooperative launch, so that the runtime
// itself refuses a grid that cannot be co-resident (hipErrorCooperativeLaunchTooLarge) instead of the kernel timing out on its bounded
// spins.  The host-side occupancy arithmetic (persistent_fits) stays in front of it — the guide (MI355X_MICROARCH.md, Correctness
// boundaries) reports the cooperative path accepting an over-size grid at some SGPR counts — and the status word stays behind it.
// The BATCHED grids are intentionally not co-resident across runs (a run that does not fit yet starts when earlier ones finish) and
// keep the plain launch.  ICRL_PLAIN_LAUNCH=1 in the environment: plain launches (A/B).
template <class K, class A>
inline hipError_t launch_coresident(K kernel, dim3 grid, dim3 block, size_t dyn_lds, hipStream_t s, A& arg) {
  static const bool plain = getenv("ICRL_PLAIN_LAUNCH") != nullptr;
  if (plain) {
    hipLaunchKernelGGL(kernel, grid, block, dyn_lds, s, arg);
    return hipGetLastError();
  }
  void* params[] = {(void*)&arg};
  return hipLaunchCooperativeKernel((const void*)kernel, grid, block, params, (unsigned)dyn_lds, s);
}
// the same with a second, scalar kernel argument
template <class K, class A>
inline hipError_t launch_coresident(K kernel, dim3 grid, dim3 block, size_t dyn_lds, hipStream_t s, A& arg, int arg2) {
  static const bool plain = getenv("ICRL_PLAIN_LAUNCH") != nullptr;
  if (plain) {
    hipLaunchKernelGGL(kernel, grid, block, dyn_lds, s, arg, arg2);
    return hipGetLastError();
  }
  void* params[] = {(void*)&arg, (void*)&arg2};
  return hipLaunchCooperativeKernel((const void*)kernel, grid, block, params, (unsigned)dyn_lds, s);
}

template <class T>
__global__ void put_args_kernel(T v, T* dst) {
  if (threadIdx.x == 0) *dst = v;
}
template <class T>
inline int put_args(const T& v, T* dst, hipStream_t s) {
  static_assert(sizeof(T) <= 3584, "argument block must fit the kernel-argument segment");
  hipLaunchKernelGGL(put_args_kernel<T>, dim3(1), dim3(64), 0, s, v, dst);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// counter-based random stream of the synthetic env (spec: oracle/synth_env.py u24())
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t u24(uint32_t key, uint32_t ctr, uint32_t comp) {
  uint32_t x = fmix32(key ^ 0x9E3779B9u);
  x = fmix32(x + ctr * 0x9E3779B1u);
  x = fmix32(x ^ (comp * 0x7FEB352Du));
  return x >> 8;
}
__host__ __device__ inline double unit_uniform(uint32_t key, uint32_t ctr, uint32_t comp) {
  return (double)u24(key, ctr, comp) / 16777216.0;
}

// ---------------------------------------------------------------------------------------------------------------
// wave-level reductions (butterfly over 64 lanes; every lane ends with the total)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}

// ---- the same reductions on the VALU only (DPP row operations + gfx950's v_permlane{16,32}_swap): ~8 instructions instead of
// six dependent ds_bpermute round trips through the LDS crossbar.  Summation order differs from the butterflies above.
__device__ __forceinline__ float xor32_sum(float v) {   // v[lane] + v[lane ^ 32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float v) {   // v[lane] + v[lane ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#define ICRL_DPP_F32(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float row_sum(float v) {     // sum over the 16 lanes of the lane's row, in every lane
  v += ICRL_DPP_F32(v, 0xB1);    // quad_perm [1,0,3,2]
  v += ICRL_DPP_F32(v, 0x4E);    // quad_perm [2,3,0,1]
  v += ICRL_DPP_F32(v, 0x141);   // row_half_mirror
  v += ICRL_DPP_F32(v, 0x140);   // row_mirror
  return v;
}
__device__ __forceinline__ float row_max(float v) {     // max over the 16 lanes of the lane's row, in every lane
  v = fmaxf(v, ICRL_DPP_F32(v, 0xB1));
  v = fmaxf(v, ICRL_DPP_F32(v, 0x4E));
  v = fmaxf(v, ICRL_DPP_F32(v, 0x141));
  v = fmaxf(v, ICRL_DPP_F32(v, 0x140));
  return v;
}
__device__ __forceinline__ float quad_rows_sum(float v) { return xor16_sum(xor32_sum(v)); }   // over the 4 lanes lane % 16
__device__ __forceinline__ float wave_sum_fast(float v) { return row_sum(quad_rows_sum(v)); }

// numpy's pairwise summation of a contiguous float64 vector (np.sum / np.mean over a 1-D array); reproduces the
// reference's reduction order exactly so that ret_rms / cost_rms match bit for bit.
__device__ inline double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// ---------------------------------------------------------------------------------------------------------------
// XCD placement of persistent workgroups that exchange granules (update kernels: ppo_common.h; batched multi-env rollout)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
// called by ONE thread of workgroup j (of M) of a run; `words`: M zeroed 8-byte words of the run.  True when all M workgroups report
// the same XCD (bounded wait: a workgroup that does not show up counts as elsewhere)
__device__ __forceinline__ bool all_on_one_xcd(unsigned long long* words, int stride_words, int j, int M) {
  const unsigned long long me = 0x100ull | (unsigned long long)xcc_id();
  __hip_atomic_store(words + (size_t)j * stride_words, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool same = true;
  for (int o = 0; o < M; ++o) {
    if (o == j) continue;
    unsigned long long v = 0;
    for (int spins = 0; spins < (1 << 18); ++spins) {
      v = __hip_atomic_load(words + (size_t)o * stride_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
    same = same && v == me;
  }
  return same;
}

// ---------------------------------------------------------------------------------------------------------------
// flat parameter layouts
// ---------------------------------------------------------------------------------------------------------------
struct PolLayout {
  int O, A, H1, H2, discrete, n;
  int log_std;                      // -1 when discrete
  int W1[3], b1[3], W2[3], b2[3];   // pi, vf, cvf
  int Wa, ba, Wv, bv, Wc, bc;
};

__host__ __device__ inline PolLayout make_pol_layout(int O, int A, int H1, int H2, int discrete) {
  PolLayout L;
  L.O = O; L.A = A; L.H1 = H1; L.H2 = H2; L.discrete = discrete;
  int off = 0;
  if (discrete) L.log_std = -1; else { L.log_std = 0; off += A; }
  for (int w = 0; w < 3; ++w) {
    L.W1[w] = off; off += H1 * O;
    L.b1[w] = off; off += H1;
    L.W2[w] = off; off += H2 * H1;
    L.b2[w] = off; off += H2;
  }
  L.Wa = off; off += A * H2;
  L.ba = off; off += A;
  L.Wv = off; off += H2;
  L.bv = off; off += 1;
  L.Wc = off; off += H2;
  L.bc = off; off += 1;
  L.n = off;
  return L;
}

struct CnLayout {
  int in, nh, H1, H2, n;
  int W0, b0, W1, b1, Wo, bo;
};

__host__ __device__ inline CnLayout make_cn_layout(int in, int nh, int H1, int H2) {
  CnLayout L;
  L.in = in; L.nh = nh; L.H1 = H1; L.H2 = (nh == 2 ? H2 : H1);
  int off = 0;
  L.W0 = off; off += H1 * in;
  L.b0 = off; off += H1;
  if (nh == 2) { L.W1 = off; off += H2 * H1; L.b1 = off; off += H2; } else { L.W1 = L.b1 = -1; }
  L.Wo = off; off += L.H2;
  L.bo = off; off += 1;
  L.n = off;
  return L;
}

// tanh through one v_exp_f32 and one v_rcp_f32: 1 - 2 / (2^(2 x log2 e) + 1); |abs error| <~ 2e-7, saturates to +-1 without
// clamps (2^big = inf -> rcp = 0; 2^-big = 0 -> rcp(1) = 1).  Used by BOTH the rollout-time forward and the training kernels
// so that old and new log-probs see the same activations.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

constexpr float LOG_SQRT_2PI_F = 0.918938533204672741780329736406f;  // log(sqrt(2*pi))
constexpr float HALF_LOG_2PI_PLUS_HALF_F = 1.418938533204672741780329736406f;  // 0.5 + 0.5*log(2*pi)

}  // namespace icrl
