// Rollout-time kernels — gfx950.
//
// One environment step of OnPolicyWithCostAlgorithm.collect_rollouts
// (ref: stable_baselines3/common/on_policy_algorithm.py:367-416) is two launches:
//
//   act_step_kernel   one workgroup (3 waves = pi | vf | cvf) per environment:
//                     policy forward (ref: policies.py:716-731, torch_layers.py:245-254, distributions.py:143-171)
//                     -> clip -> synthetic env step + auto-reset (spec: oracle/synth_env.py)
//                     -> cost_function(previous raw obs, clipped action) (ref: icrl/constraint_net.py:121-130,258-299;
//                        vec_cost_wrapper.py:51-66) -> the buffer fields that do not need the normaliser
//                        (ref: buffers.py:554-592).
//   norm_step_kernel  one workgroup for all environments: running-moment merge of obs / discounted reward return /
//                     discounted cost return in float64 with numpy's reduction order, normalise + clip, remaining
//                     buffer fields (ref: vec_normalize.py:81-123,220-261; running_mean_std.py:19-39).
//
// The same device functions back the fine-grained C-ABI entry points (icrl_policy_forward, icrl_cost_mlp_forward,
// icrl_synth_env_step, icrl_vecnorm_step) that the Python VecEnv / ConstraintNet classes call one at a time.
//
// Weights are read from the transposed copies ([in][out]) written by *_prepare so that the 64 lanes of a wave
// (one lane per output unit) load 256 contiguous bytes per input index.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "generic.h"

namespace icrl {

// =================================================================================================================
// transposed weight copies
// =================================================================================================================
__global__ void policy_transpose_kernel(PolLayout L, const float* __restrict__ p, float* __restrict__ pt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L.n; i += gridDim.x * blockDim.x) {
    int dst = i;
    for (int w = 0; w < 3; ++w) {
      if (i >= L.W1[w] && i < L.b1[w]) { int r = i - L.W1[w]; int j = r / L.O, k = r % L.O; dst = L.W1[w] + k * L.H1 + j; }
      if (i >= L.W2[w] && i < L.b2[w]) { int r = i - L.W2[w]; int j = r / L.H1, k = r % L.H1; dst = L.W2[w] + k * L.H2 + j; }
    }
    if (i >= L.Wa && i < L.ba) { int r = i - L.Wa; int a = r / L.H2, j = r % L.H2; dst = L.Wa + j * L.A + a; }
    pt[dst] = p[i];
  }
}

__global__ void costnet_transpose_kernel(CnLayout L, const float* __restrict__ p, float* __restrict__ pt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L.n; i += gridDim.x * blockDim.x) {
    int dst = i;
    if (i >= L.W0 && i < L.b0) { int r = i - L.W0; int j = r / L.in, k = r % L.in; dst = L.W0 + k * L.H1 + j; }
    if (L.nh == 2 && i >= L.W1 && i < L.b1) { int r = i - L.W1; int j = r / L.H1, k = r % L.H1; dst = L.W1 + k * L.H2 + j; }
    pt[dst] = p[i];
  }
}

// =================================================================================================================
// device building blocks (called by a 192-thread workgroup handling ONE environment)
// =================================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ActShared {
  alignas(16) double s_old[MAX_OBS];   // raw observation before the step (== VecCostWrapper.previous_obs)
  alignas(16) double s_new[MAX_OBS];   // raw observation after the step (post auto-reset)
  alignas(16) float x[MAX_OBS];        // normalised observation, float32 (policy input)
  alignas(16) float h[3][MAX_H];
  alignas(16) float g[3][MAX_H];
  alignas(16) float cx[MAX_CN_IN];     // cost-net input
  alignas(16) float ch[2][MAX_H];
  float act_raw[MAX_ACT];
  float act_clip[MAX_ACT];
  float scal[4];           // v_r, v_c, log_prob, (unused)
};

// One lane's slice of the three MLPs, held in registers: lane j of wave w (pi | vf | cvf) owns column j of W1^T and
// W2^T of network w; lanes < A of wave 0 own a column of the action head, the value waves one head weight per lane.
// All loads are issued back to back (one memory latency for the whole network instead of one per input index); a
// persistent kernel loads them ONCE for all its steps.
// ONE register image per wave for both roles: a wave is either a policy wave (pi / vf / cvf) or the cost-net wave, never both,
// but the compiler cannot know that and would keep two full weight sets live in every wave (at AntWall widths 263 + 232
// values: 468 registers spilled to scratch, the cost wave alone 50 k cycles per step).  The two roles' arrays of equal type
// share storage through anonymous unions (same element type: plain aliases, no type punning).
template <int OCT, int CIT>
struct WaveRegs {
  static constexpr int NA = 16 * (OCT > CIT ? OCT : CIT);
  union { float w1[NA]; float c0[NA]; };          // policy: W1 column of hidden unit `lane` (16 OCT used) | cost net: W0 column (16 CIT)
  union { float w2[MAX_H]; float c1[MAX_H]; };    // second layer column
  float wh[MAX_H];   // wave 0: head column of action `lane`; waves 1/2: wh[0] = value-head weight of hidden unit `lane`
  union { float b1; float cb0; };
  union { float b2; float cb1; };
  union { float bh; float cbo; };
  union { float ls; float co; };
  float sd, lsd, i2v;   // Gaussian head constants of action `lane`: exp(log_std), log(exp(log_std)), 1 / (2 sd^2) denominators
  int sel[(16 * (CIT > 0 ? CIT : 1) + WAVE - 1) / WAVE];   // cost net: select_dim entries this lane prepares
};
template <int OCT> using PolRegs = WaveRegs<OCT, 0>;
template <int CIT> using CnRegs = WaveRegs<0, CIT>;

template <int OCT, class Regs>
__device__ __forceinline__ void load_pol_regs(const PolLayout& L, const float* __restrict__ PT, Regs& R) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j1 = lane < L.H1 ? lane : 0, j2 = lane < L.H2 ? lane : 0;
#pragma unroll
  for (int k = 0; k < 16 * OCT; ++k) R.w1[k] = (w < 3 && k < L.O) ? PT[L.W1[w] + k * L.H1 + j1] : 0.f;
#pragma unroll
  for (int k = 0; k < MAX_H; ++k) R.w2[k] = (w < 3 && k < L.H1) ? PT[L.W2[w] + k * L.H2 + j2] : 0.f;
  R.b1 = w < 3 ? PT[L.b1[w] + j1] : 0.f;
  R.b2 = w < 3 ? PT[L.b2[w] + j2] : 0.f;
  R.bh = 0.f; R.ls = 0.f; R.sd = 1.f; R.lsd = 0.f; R.i2v = 2.f;
  if (w == 0) {
    const int a = lane < L.A ? lane : 0;
#pragma unroll
    for (int j = 0; j < MAX_H; ++j) R.wh[j] = j < L.H2 ? PT[L.Wa + j * L.A + a] : 0.f;
    R.bh = PT[L.ba + a];
    if (!L.discrete) R.ls = PT[L.log_std + a];
    // constants of the whole rollout: evaluated once here instead of once per env step (same expressions as before)
    R.sd = expf(R.ls); R.lsd = logf(R.sd); R.i2v = 2.f * (R.sd * R.sd);
  } else if (w < 3) {
#pragma unroll
    for (int j = 1; j < MAX_H; ++j) R.wh[j] = 0.f;
    R.wh[0] = PT[(w == 1 ? L.Wv : L.Wc) + j2];
    R.bh = PT[w == 1 ? L.bv : L.bc];
  }
}

// three tanh MLPs + heads for the observation in sh.x.  Must be called by all threads of the workgroup (waves >= 3 only
// take part in the barriers).
template <int OCT, class Regs>
__device__ __forceinline__ void policy_forward_block(const PolLayout& L, const Regs& R, ActShared& sh,
                                                     const float* noise_row, int deterministic,
                                                     const float* alow, const float* ahigh,
                                                     const float* given = nullptr /* evaluate_actions: actions to score */) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // activations are pulled from LDS as whole float4 rows first (broadcast reads), then consumed from registers: one LDS
  // latency per layer instead of one per input.  Pad weights are zero, so no bounds test is needed in the FMA chains.
  if (w < 3) {
    f32x4 xs[4 * OCT];
#pragma unroll
    for (int i = 0; i < 4 * OCT; ++i) xs[i] = *reinterpret_cast<const f32x4*>(&sh.x[4 * i]);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16 * OCT; ++k) acc = fmaf(R.w1[k], xs[k >> 2][k & 3], acc);
    sh.h[w][lane] = lane < L.H1 ? fast_tanh(acc + R.b1) : 0.f;   // pad lanes store 0 (0 * garbage would poison the next layer)
  }
  __syncthreads();
  if (w < 3) {
    f32x4 hs[MAX_H / 4];
#pragma unroll
    for (int i = 0; i < MAX_H / 4; ++i) hs[i] = *reinterpret_cast<const f32x4*>(&sh.h[w][4 * i]);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_H; ++k) acc = fmaf(R.w2[k], hs[k >> 2][k & 3], acc);
    sh.g[w][lane] = lane < L.H2 ? fast_tanh(acc + R.b2) : 0.f;
  }
  __syncthreads();
  if (w == 0) {
    float mean = 0.f;
    {
      f32x4 gs[MAX_H / 4];
#pragma unroll
      for (int i = 0; i < MAX_H / 4; ++i) gs[i] = *reinterpret_cast<const f32x4*>(&sh.g[0][4 * i]);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < MAX_H; ++j) acc = fmaf(R.wh[j], gs[j >> 2][j & 3], acc);
      mean = acc + R.bh;
    }
    if (!L.discrete) {
      float lp = 0.f, entl = 0.f;
      if (lane < L.A) {
        const float std = R.sd;
        float act = mean;
        if (given != nullptr) act = given[lane];
        else if (!deterministic) act = mean + noise_row[lane] * std;   // Normal.rsample: loc + eps * scale
        const float diff = act - mean;
        // Normal.log_prob: -((x - mu)^2) / (2 var) - log(std) - log(sqrt(2 pi))
        lp = -(diff * diff) / R.i2v - R.lsd - LOG_SQRT_2PI_F;
        entl = HALF_LOG_2PI_PLUS_HALF_F + R.lsd;
        sh.act_raw[lane] = act;
        float c = act;
        if (alow != nullptr && ahigh != nullptr) c = fminf(fmaxf(act, alow[lane]), ahigh[lane]);
        sh.act_clip[lane] = c;
      }
      lp = wave_sum_fast(lp);
      // entropy of the diagonal Gaussian: sum_a 0.5 + 0.5 log(2 pi) + log sigma_a
      const float ent = wave_sum_fast(entl);
      if (lane == 0) { sh.scal[2] = lp; sh.scal[3] = ent; }
    } else {
      // Categorical(logits): log-softmax, inverse-CDF sample on the injected uniform (spec: oracle/nets.py forward)
      float m = wave_max(lane < L.A ? mean : -INFINITY);
      float e = lane < L.A ? expf(mean - m) : 0.f;
      float lse = m + logf(wave_sum(e));
      float logp = mean - lse;
      float p = lane < L.A ? expf(logp) : 0.f;
      float cdf = 0.f;
      int action = 0;
      if (given != nullptr) {
        action = (int)given[0];
      } else if (deterministic) {
        float best = wave_max(lane < L.A ? p : -1.f);
        unsigned long long ball = __ballot(lane < L.A && p == best);
        action = __ffsll((long long)ball) - 1;
      } else {
        const float u = noise_row[0];
        int cnt = 0;
        for (int a = 0; a < L.A; ++a) {     // sequential inclusive prefix sum, like th.cumsum
          cdf += __shfl(p, a, 64);
          cnt += (u >= cdf) ? 1 : 0;
        }
        action = cnt < L.A - 1 ? cnt : L.A - 1;
      }
      const float lp = __shfl(logp, action, 64);
      const float ent = -wave_sum(lane < L.A ? logp * p : 0.f);
      if (lane == 0) { sh.scal[2] = lp; sh.scal[3] = ent; sh.act_raw[0] = (float)action; sh.act_clip[0] = (float)action; }
    }
  } else if (w < 3) {
    float part = lane < L.H2 ? R.wh[0] * sh.g[w][lane] : 0.f;
    part = wave_sum_fast(part);
    if (lane == 0) sh.scal[w - 1] = part + R.bh;
  }
}

// cost-net weights of one lane (lane j <-> hidden unit j), loaded with independent loads before they are needed (WaveRegs c0 / c1 / ...)
template <int CIT, class Regs>
__device__ __forceinline__ void load_cn_regs(const icrl_costnet_t& cn, const CnLayout& L, Regs& R) {
  const int lane = threadIdx.x & 63;
  const float* PT = cn.params_t;
  const int j1 = lane < L.H1 ? lane : 0, j2 = lane < L.H2 ? lane : 0;
#pragma unroll
  for (int k = 0; k < 16 * CIT; ++k) R.c0[k] = k < L.in ? PT[L.W0 + k * L.H1 + j1] : 0.f;
#pragma unroll
  for (int k = 0; k < MAX_H; ++k) R.c1[k] = (L.nh == 2 && k < L.H1) ? PT[L.W1 + k * L.H2 + j2] : 0.f;
  R.cb0 = PT[L.b0 + j1];
  R.cb1 = L.nh == 2 ? PT[L.b1 + j2] : 0.f;
  R.co = PT[L.Wo + j2];
  R.cbo = PT[L.bo];
#pragma unroll
  for (int i = 0; i < (16 * CIT + WAVE - 1) / WAVE; ++i) { const int idx = lane + i * WAVE; R.sel[i] = idx < L.in ? cn.select_dim[idx] : -1; }
}

// cost = 1 - sigmoid(ReLU-MLP(prepare(obs, acs))); called by ONE wave. obs_row: float64 raw obs (LDS or global),
// acs_row: float32 actions.  cx: >= 16*CIT floats (16-byte aligned), ch: [2][MAX_H].  Returns the cost in every lane.
template <int CIT, class Regs>
__device__ __forceinline__ float cost_forward_wave(const icrl_costnet_t& cn, const CnLayout& L, const Regs& R,
                                                   const double* obs_row, const float* acs_row, float* cx, float (*ch)[MAX_H],
                                                   int mode = 0 /* 0: cost = 1 - zeta; 1: zeta; 2: log(zeta + eps) */) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < (16 * CIT + WAVE - 1) / WAVE; ++i) {
    const int idx = lane + i * WAVE;
    const int sel = R.sel[i];
    float v = 0.f;
    if (sel >= 0) {
      if (sel < cn.obs_dim) {
        double o = obs_row[sel];
        if (cn.obs_mean != nullptr && cn.obs_var != nullptr) o = (o - cn.obs_mean[sel]) / sqrt(cn.obs_var[sel] + cn.eps);
        if (cn.clip_obs >= 0.0) o = fmin(fmax(o, -cn.clip_obs), cn.clip_obs);
        v = (float)o;
      } else {
        const int a = sel - cn.obs_dim;
        float x;
        if (cn.is_discrete) x = ((int)acs_row[0] == a) ? 1.f : 0.f;
        else x = acs_row[a];
        if (cn.action_low != nullptr && cn.action_high != nullptr) x = fminf(fmaxf(x, cn.action_low[a]), cn.action_high[a]);
        v = x;
      }
    }
    if (idx < 16 * CIT) cx[idx] = v;       // pad entries are written as 0
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS writes of this wave visible to its own reads
  {
    f32x4 xs[4 * CIT];
#pragma unroll
    for (int i = 0; i < 4 * CIT; ++i) xs[i] = *reinterpret_cast<const f32x4*>(&cx[4 * i]);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16 * CIT; ++k) acc = fmaf(R.c0[k], xs[k >> 2][k & 3], acc);
    ch[0][lane] = lane < L.H1 ? fmaxf(acc + R.cb0, 0.f) : 0.f;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int last = 0;
  if (L.nh == 2) {
    f32x4 hs[MAX_H / 4];
#pragma unroll
    for (int i = 0; i < MAX_H / 4; ++i) hs[i] = *reinterpret_cast<const f32x4*>(&ch[0][4 * i]);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_H; ++k) acc = fmaf(R.c1[k], hs[k >> 2][k & 3], acc);
    ch[1][lane] = lane < L.H2 ? fmaxf(acc + R.cb1, 0.f) : 0.f;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    last = 1;
  }
  float part = lane < L.H2 ? R.co * ch[last][lane] : 0.f;
  const float z = wave_sum_fast(part) + R.cbo;
  const float zeta = 1.f / (1.f + expf(-z));
  if (mode == 1) return zeta;
  if (mode == 2) return logf(zeta + (float)cn.eps);
  return 1.f - zeta;
}

// synthetic env step for env n; called by ONE wave.  act: float32 clipped actions (LDS or global).
// Returns reward / done in every lane; writes env.s, t_ep, step_count.  s_old: float64 previous state (LDS copy).
// observation an env (re)starts from: the synthetic envs draw it from their stream, LapGridWorld always starts in cell 0
__device__ __forceinline__ double env_reset_value(const icrl_env_t& e, uint32_t key, uint32_t ctr, int i) {
  if (e.reward_form >= 2) return -1.0;   // ((0 - 0) * 2) / 40 - 1
  return (unit_uniform(key, ctr, (uint32_t)(e.obs_dim + i)) - 0.5) * 0.2;
}

// LapGridWorld / ConstrainedLapGridWorld (ref: custom_envs/custom_envs/envs/lap_grid_world.py:62-119,197-240; spec:
// oracle/lap_grid.py): 40 cells, coins worth 3 at 5/15/25/35, action 0 forward / 1 backward, 200-step episodes.
// The state IS the observation 2*pos/40 - 1 (float64), so the cell index is recovered exactly by rounding.
__device__ __forceinline__ void lap_grid_step_wave(const icrl_env_t& e, int n, const double* s_old, const float* act,
                                                   uint32_t& ctr_io, int& tep_io, double* s_new_lds, double& reward,
                                                   int& done) {
  const int lane = threadIdx.x & 63;
  const int pos_old = (int)llrint((s_old[0] + 1.0) * 20.0);
  const int ac = (int)act[0];
  int pos = pos_old, d = 0;
  double rew;
  if (ac == 0) {
    pos = pos_old + 1 == 40 ? 0 : pos_old + 1;
    rew = (pos % 10 == 5) ? 3.0 : 0.0;
  } else if (e.reward_form == 3) {   // constrained: moving backward is penalised and ends the episode
    rew = -1.0;
    d = 1;
  } else {
    pos = pos_old == 0 ? 39 : pos_old - 1;
    rew = (pos % 10 == 5) ? 3.0 : 0.0;
  }
  const int tep = tep_io + 1;
  if (tep >= e.max_steps) d = 1;
  if (d) pos = 0;
  const double v = ((double)pos * 2.0) / 40.0 - 1.0;
  tep_io = d ? 0 : tep;
  ctr_io = ctr_io + 1u;
  if (lane == 0) {
    e.s[n] = v;
    if (s_new_lds != nullptr) s_new_lds[0] = v;
    e.t_ep[n] = tep_io;
    e.step_count[n] = ctr_io;
  }
  reward = rew;
  done = d;
}

__device__ __forceinline__ void env_step_wave(const icrl_env_t& e, int n, const double* s_old, const float* act,
                                              uint32_t key, uint32_t& ctr_io, int& tep_io, double* s_new_lds,
                                              double& reward, int& done) {
  if (e.reward_form >= 2) {
    lap_grid_step_wave(e, n, s_old, act, ctr_io, tep_io, s_new_lds, reward, done);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int O = e.obs_dim, A = e.act_dim;
  double a[MAX_ACT];
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < MAX_ACT; ++j) {
    a[j] = 0.0;
    if (j < A) {
      a[j] = (double)act[j];
      if (e.broken && j >= 4) a[j] = 0.0;
      sq = sq + a[j] * a[j];
    }
  }
  const uint32_t ctr = ctr_io;
  double ns[2];  // this lane's components i = lane, lane + 64
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = lane + r * WAVE;
    ns[r] = 0.0;
    if (i < O) {
      double acc = 0.99 * s_old[i];
      const double* Bi = e.B + (size_t)i * A;
#pragma unroll
      for (int j = 0; j < MAX_ACT; ++j)
        if (j < A) acc = acc + Bi[j] * a[j];
      const double eps = (unit_uniform(key, ctr, (uint32_t)i) - 0.5) * 3.4641016151377544;
      ns[r] = acc + 0.01 * eps;
    }
  }
  const double n0 = __shfl(ns[0], 0, 64);
  const double n1 = __shfl(ns[0], 1, 64);
  double rew;
  if (e.reward_form == 0) rew = fabs(n0 - s_old[0]) / 0.05 - 0.1 * sq;
  else rew = (sqrt(n0 * n0 + n1 * n1) + 1.0) - 0.5 * sq;
  int d = 0;
  if (e.wall_terminate && n0 <= -3.0) { rew = 0.0; d = 1; }
  const int tep = tep_io + 1;
  if (tep >= e.max_steps) d = 1;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = lane + r * WAVE;
    if (i < O) {
      double v = ns[r];
      if (d) v = env_reset_value(e, key, ctr + 1u, i);   // auto-reset draw
      e.s[(size_t)n * O + i] = v;
      if (s_new_lds != nullptr) s_new_lds[i] = v;
    }
  }
  tep_io = d ? 0 : tep;
  ctr_io = ctr + 1u;
  if (lane == 0) {
    e.t_ep[n] = tep_io;
    e.step_count[n] = ctr_io;
  }
  reward = rew;
  done = d;
}

// =================================================================================================================
// kernel A: policy forward + env step + cost + pre-normaliser buffer fields, one workgroup per env
// =================================================================================================================
struct ActStepArgs {
  icrl_env_t env;
  icrl_costnet_t cn;
  icrl_buffer_t buf;
  icrl_agent_t ag;
  PolLayout pl;
  CnLayout cl;
  const float* PT;
  const float* noise;   // [T,N,act] (or [T,N] uniforms when discrete)
  const float* alow;
  const float* ahigh;
  int has_cn;
};

template <int OCT, int CIT>
__global__ void __launch_bounds__(256) act_step_kernel(ActStepArgs a, int t) {
  __shared__ ActShared sh;
  WaveRegs<OCT, CIT> R;                // one image: policy weights in waves 0..2, cost-net weights in wave 3
  WaveRegs<OCT, CIT>& C = R;
  load_pol_regs<OCT>(a.pl, a.PT, R);   // every weight load of the step is in flight before anything waits
  if (threadIdx.x >= 192 && a.has_cn) load_cn_regs<CIT>(a.cn, a.cl, C);   // wave 3 = cost net
  const int n = blockIdx.x;
  const uint32_t e_key = a.env.key[n];
  uint32_t e_ctr = a.env.step_count[n];
  int e_tep = a.env.t_ep[n];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = a.pl.O, A = a.pl.A, N = a.env.n_envs;
  const int AS = a.buf.act_store;
  for (int i = tid; i < MAX_OBS; i += 256) {
    sh.x[i] = i < O ? (float)a.ag.last_obs[(size_t)n * O + i] : 0.f;     // preprocess_obs: .float(); pad = 0
    if (i < O) sh.s_old[i] = a.env.s[(size_t)n * O + i];
  }
  __syncthreads();
  const size_t tn = (size_t)t * N + n;
  const float* noise_row = a.noise + tn * (a.pl.discrete ? 1 : A);
  policy_forward_block<OCT>(a.pl, R, sh, noise_row, 0, a.alow, a.ahigh);
  __syncthreads();
  if (w == 0) {
    double rew; int done;
    env_step_wave(a.env, n, sh.s_old, sh.act_clip, e_key, e_ctr, e_tep, sh.s_new, rew, done);
    float* nob = a.buf.new_orig_observations + tn * O;
    for (int i = lane; i < O; i += WAVE) nob[i] = (float)sh.s_new[i];
    if (lane == 0) { a.ag.raw_rew[n] = rew; a.ag.dones[n] = (uint8_t)done; }
  } else if (w == 3) {
    float cost = 0.f;
    if (a.has_cn) cost = cost_forward_wave<CIT>(a.cn, a.cl, C, sh.s_old, sh.act_clip, sh.cx, sh.ch);
    if (lane == 0) { a.ag.raw_cost[n] = cost; a.buf.orig_costs[tn] = cost; }
  } else if (w == 2) {
    float* ob = a.buf.observations + tn * O;
    float* oob = a.buf.orig_observations + tn * O;
    for (int i = lane; i < O; i += WAVE) { ob[i] = sh.x[i]; oob[i] = (float)sh.s_old[i]; }
    if (lane < AS) a.buf.actions[tn * AS + lane] = sh.act_raw[lane];
    if (lane < A && !a.pl.discrete) a.ag.act_clipped[(size_t)n * A + lane] = sh.act_clip[lane];
    if (lane == 0) {
      a.buf.dones[tn] = (float)a.ag.last_dones[n];
      a.buf.reward_values[tn] = sh.scal[0];
      a.buf.cost_values[tn] = sh.scal[1];
      a.buf.log_probs[tn] = sh.scal[2];
      a.ag.last_v_r[n] = sh.scal[0];
      a.ag.last_v_c[n] = sh.scal[1];
    }
  }
}

// act_step_kernel for policies / constraint nets the one-workgroup-per-env image does not hold (generic-shape path): the policy
// forward (policy_generic_kernel: actions, values, log-probs straight into row t of the buffer, clipped actions into ag.act_clipped)
// and the cost (cn_cost_rows_kernel: ag.raw_cost) have run as launches of their own on the same stream; this kernel is the rest
// of the step for env n — env step, the pre-normaliser buffer fields, the values of this forward for the GAE bootstrap.  One wave.
struct GenStepArgs {
  icrl_env_t env;
  icrl_buffer_t buf;
  icrl_agent_t ag;
  int has_cn;
};

__global__ void __launch_bounds__(64) act_step_generic_kernel(GenStepArgs a, int t) {
  __shared__ double s_old[MAX_OBS], s_new[MAX_OBS];
  __shared__ float act_clip[MAX_ACT];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int O = a.env.obs_dim, N = a.env.n_envs, AS = a.buf.act_store;
  const uint32_t e_key = a.env.key[n];
  uint32_t e_ctr = a.env.step_count[n];
  int e_tep = a.env.t_ep[n];
  const size_t tn = (size_t)t * N + n;
  float* ob = a.buf.observations + tn * O;
  float* oob = a.buf.orig_observations + tn * O;
  for (int i = lane; i < O; i += WAVE) {
    const double s = a.env.s[(size_t)n * O + i];
    s_old[i] = s;
    oob[i] = (float)s;
    ob[i] = (float)a.ag.last_obs[(size_t)n * O + i];      // preprocess_obs: .float()
  }
  if (lane < AS) act_clip[lane] = a.ag.act_clipped[(size_t)n * AS + lane];
  __syncthreads();
  double rew; int done;
  env_step_wave(a.env, n, s_old, act_clip, e_key, e_ctr, e_tep, s_new, rew, done);
  float* nob = a.buf.new_orig_observations + tn * O;
  for (int i = lane; i < O; i += WAVE) nob[i] = (float)s_new[i];
  if (lane == 0) {
    a.ag.raw_rew[n] = rew; a.ag.dones[n] = (uint8_t)done;
    const float cost = a.has_cn ? a.ag.raw_cost[n] : 0.f;
    a.ag.raw_cost[n] = cost; a.buf.orig_costs[tn] = cost;
    a.buf.dones[tn] = (float)a.ag.last_dones[n];
    a.ag.last_v_r[n] = a.buf.reward_values[tn];
    a.ag.last_v_c[n] = a.buf.cost_values[tn];
  }
}

// =================================================================================================================
// kernel B: VecNormalizeWithCost.step_wait for all envs, one workgroup
// =================================================================================================================
struct NormStepArgs {
  icrl_norm_t nm;
  const double* raw_obs;   // [N,O]
  const double* raw_rew;   // [N]
  const float* raw_cost;   // [N] or NULL
  const uint8_t* dones;    // [N]
  int N, O;
  double* obs_out;         // [N,O] float64 or NULL
  double* rew_out;         // [N] or NULL
  double* cost_out;        // [N] or NULL
  float* obs_f32;          // buffer.new_observations[t] or NULL
  float* rew_f32;          // buffer.rewards[t] or NULL
  float* cost_f32;         // buffer.costs[t] or NULL
  uint8_t* last_dones;     // [N] or NULL: receives dones
};

// Chan merge with the reference's operation order (running_mean_std.py:25-39).
__device__ __forceinline__ void chan_merge(double& mean, double& var, double count, double b_mean, double b_var,
                                           double b_count) {
  const double delta = b_mean - mean;
  const double tot = count + b_count;
  const double new_mean = mean + delta * b_count / tot;
  const double m_a = var * count;
  const double m_b = b_var * b_count;
  const double m_2 = m_a + m_b + (delta * delta) * count * b_count / (count + b_count);
  mean = new_mean;
  var = m_2 / (count + b_count);
}

// ---- numpy-ordered 1-D sum, parallelised without changing the rounding order -------------------------------------
// np.sum of a contiguous float64 vector splits recursively (n2 = n/2 rounded down to a multiple of 8) until blocks of
// <= 128 elements, sums each block with 8 strided accumulators + a fixed combine tree + sequential tail, and adds the
// block results back up the recursion.  Blocks and accumulators are independent, so (block, accumulator) pairs map to
// threads; only the tiny combine runs on one thread.
constexpr int NORM_MAX_N = 4096;                 // envs handled by the single-workgroup normaliser (and per GPU)
constexpr int WIDE_MAX_N = 1024;                 // rollout_wide_kernel's static buffers; beyond: rollout_multi_kernel
constexpr int NORM_MAX_LEAVES = NORM_MAX_N / 64 + 2;
constexpr int NORM_CHUNK = 4096;                 // doubles of raw obs staged in LDS per pass

struct NormShared {
  double chunk[NORM_CHUNK];
  double vec[2][NORM_MAX_N];                     // ret / cost_ret (then their squared deviations)
  double part[2][NORM_MAX_LEAVES][8];
  double leaf[2][NORM_MAX_LEAVES];
  double total[2];
  int leaf_off[NORM_MAX_LEAVES], leaf_len[NORM_MAX_LEAVES], n_leaves;
};

__device__ inline double np_combine(int l, const double* leaf, int& idx) {
  if (l <= 128) return leaf[idx++];
  int n2 = l / 2;
  n2 -= n2 % 8;
  const double a = np_combine(n2, leaf, idx);
  const double b = np_combine(l - n2, leaf, idx);
  return a + b;
}

// sums sh.vec[0][0:n] and (if two) sh.vec[1][0:n] into sh.total[]; all threads of the block must call.
__device__ inline void block_np_sum(NormShared& sh, int n, bool two) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (tid == 0) {
    int so[16], sl[16], sp = 1, cnt = 0;
    so[0] = 0; sl[0] = n;
    while (sp) {
      --sp;
      const int o = so[sp], l = sl[sp];
      if (l <= 128) { sh.leaf_off[cnt] = o; sh.leaf_len[cnt] = l; ++cnt; }
      else { int n2 = l / 2; n2 -= n2 % 8; so[sp] = o + n2; sl[sp] = l - n2; ++sp; so[sp] = o; sl[sp] = n2; ++sp; }
    }
    sh.n_leaves = cnt;
  }
  __syncthreads();
  const int nl = sh.n_leaves, nv = two ? 2 : 1;
  for (int job = tid; job < nv * nl * 8; job += nt) {
    const int v = job / (nl * 8), L = (job / 8) % nl, k = job % 8;
    const int o = sh.leaf_off[L], l = sh.leaf_len[L];
    const double* a = sh.vec[v] + o;
    if (l >= 8) {
      double r = a[k];
      for (int i = 8; i < l - (l % 8); i += 8) r += a[i + k];
      sh.part[v][L][k] = r;
    }
  }
  __syncthreads();
  for (int job = tid; job < nv * nl; job += nt) {
    const int v = job / nl, L = job % nl;
    const int o = sh.leaf_off[L], l = sh.leaf_len[L];
    const double* a = sh.vec[v] + o;
    double res;
    if (l < 8) {
      res = 0.0;
      for (int i = 0; i < l; ++i) res += a[i];
    } else {
      const double* r = sh.part[v][L];
      res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (int i = l - (l % 8); i < l; ++i) res += a[i];
    }
    sh.leaf[v][L] = res;
  }
  __syncthreads();
  if (tid < nv) { int idx = 0; sh.total[tid] = np_combine(n, sh.leaf[tid], idx); }
  __syncthreads();
}

__global__ void __launch_bounds__(1024) norm_step_kernel(NormStepArgs a) {
  __shared__ NormShared sh;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = a.N, O = a.O;
  const icrl_norm_t& nm = a.nm;
  const bool has_cost = a.raw_cost != nullptr;
  if (nm.training) {
    // ---- discounted returns (vec_normalize.py:102-105, 245-248), then np.mean / np.var of them
    for (int n = tid; n < N; n += nt) {
      const double r = nm.ret[n] * nm.reward_gamma + a.raw_rew[n];
      nm.ret[n] = r; sh.vec[0][n] = r;
      if (has_cost) { const double c = nm.cost_ret[n] * nm.cost_gamma + (double)a.raw_cost[n]; nm.cost_ret[n] = c; sh.vec[1][n] = c; }
    }
    __syncthreads();
    block_np_sum(sh, N, has_cost);
    const double bm_r = sh.total[0] / (double)N, bm_c = sh.total[1] / (double)N;
    __syncthreads();
    for (int n = tid; n < N; n += nt) {
      const double d = sh.vec[0][n] - bm_r; sh.vec[0][n] = d * d;
      if (has_cost) { const double e = sh.vec[1][n] - bm_c; sh.vec[1][n] = e * e; }
    }
    __syncthreads();
    block_np_sum(sh, N, has_cost);
    if (tid == 0) {
      double m = nm.ret_stats[0], v = nm.ret_stats[1];
      chan_merge(m, v, nm.ret_stats[2], bm_r, sh.total[0] / (double)N, (double)N);
      nm.ret_stats[0] = m; nm.ret_stats[1] = v; nm.ret_stats[2] = (double)N + nm.ret_stats[2];
      if (has_cost) {
        m = nm.cost_stats[0]; v = nm.cost_stats[1];
        chan_merge(m, v, nm.cost_stats[2], bm_c, sh.total[1] / (double)N, (double)N);
        nm.cost_stats[0] = m; nm.cost_stats[1] = v; nm.cost_stats[2] = (double)N + nm.cost_stats[2];
      }
    }
    // ---- obs_rms.update: numpy's axis-0 reduction adds the rows in order; rows are staged through LDS in chunks
    // (coalesced, all threads) and one thread per column accumulates them sequentially.
    const int rows_per = NORM_CHUNK / O;
    double sum = 0.0, bm = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      sum = 0.0;
      for (int r0 = 0; r0 < N; r0 += rows_per) {
        const int rows = (N - r0) < rows_per ? (N - r0) : rows_per;
        __syncthreads();
        for (int i = tid; i < rows * O; i += nt) sh.chunk[i] = a.raw_obs[(size_t)r0 * O + i];
        __syncthreads();
        if (tid < O) {
          if (pass == 0) for (int r = 0; r < rows; ++r) sum += sh.chunk[r * O + tid];
          else for (int r = 0; r < rows; ++r) { const double d = sh.chunk[r * O + tid] - bm; sum += d * d; }
        }
      }
      if (pass == 0) bm = sum / (double)N;
    }
    if (tid < O) {
      double m = nm.obs_mean[tid], v = nm.obs_var[tid];
      chan_merge(m, v, nm.obs_count[0], bm, sum / (double)N, (double)N);
      nm.obs_mean[tid] = m; nm.obs_var[tid] = v;
    }
    __syncthreads();
    if (tid == 0) nm.obs_count[0] = (double)N + nm.obs_count[0];
    __threadfence_block();
    __syncthreads();
  }
  // ---- normalise + clip (vec_normalize.py:107-123, 254-261)
  for (int idx = tid; idx < N * O; idx += nt) {
    const int j = idx % O;
    double o = a.raw_obs[idx];
    if (nm.norm_obs) o = fmin(fmax((o - nm.obs_mean[j]) / sqrt(nm.obs_var[j] + nm.epsilon), -nm.clip_obs), nm.clip_obs);
    if (a.obs_out) a.obs_out[idx] = o;
    if (a.obs_f32) a.obs_f32[idx] = (float)o;
  }
  const double rden = sqrt(nm.ret_stats[1] + nm.epsilon);
  const double cden = sqrt(nm.cost_stats[1] + nm.epsilon);
  for (int n = tid; n < N; n += nt) {
    double r = a.raw_rew[n];
    if (nm.norm_reward) r = fmin(fmax(r / rden, -nm.clip_reward), nm.clip_reward);
    if (a.rew_out) a.rew_out[n] = r;
    if (a.rew_f32) a.rew_f32[n] = (float)r;
    const int d = a.dones[n];
    if (d) nm.ret[n] = 0.0;
    if (has_cost) {
      double c = (double)a.raw_cost[n];
      if (nm.norm_cost) c = fmin(fmax(c / cden, -nm.clip_cost), nm.clip_cost);
      if (a.cost_out) a.cost_out[n] = c;
      if (a.cost_f32) a.cost_f32[n] = (float)c;
      if (d) nm.cost_ret[n] = 0.0;
    }
    if (a.last_dones) a.last_dones[n] = (uint8_t)d;
  }
}

// ---- small-problem variant (N <= 128, N*O <= NORM_CHUNK: e.g. HCWithPos with 64 envs): every input is pulled into LDS /
// registers with ONE round of global loads, the three statistics are computed concurrently by different waves
// (observation columns: threads < O; reward return: wave 15; cost return: wave 14 — for N <= 128 numpy's pairwise sum is a
// single 8-accumulator leaf, which 8 lanes reproduce exactly), one barrier, then everything is normalised from LDS.
// numpy's pairwise sum for n in {128, 256, 512, 1024} (leaves of exactly 128, a balanced tree above them), evaluated by one
// wave: lane 8 L + k owns accumulator k of leaf L (a 16-add chain instead of n adds in every lane); the xor butterflies
// reproduce the leaf's ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the recursion's left + right (float64 addition is commutative
// bit for bit, so only the association matters).  Result in every lane.  Other n: np_pairwise_sum (common.h).
__device__ __forceinline__ double np_pairwise_sum_wave(const double* a, int n) {
  const int leaves = n >> 7;
  if ((n & 127) != 0 || (leaves & (leaves - 1)) != 0 || leaves > 8) return np_pairwise_sum(a, n);
  const int lane = threadIdx.x & 63;
  const int L = lane >> 3, k = lane & 7;
  const double* p = a + 128 * (L < leaves ? L : 0) + k;
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = p[8 * i];
  double r = v[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) r += v[i];
  r += __shfl_xor(r, 1, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 4, 64);
  if (leaves >= 2) r += __shfl_xor(r, 8, 64);
  if (leaves >= 4) r += __shfl_xor(r, 16, 64);
  if (leaves >= 8) r += __shfl_xor(r, 32, 64);
  return __shfl(r, 0, 64);
}

__device__ __forceinline__ double np_leaf_sum_wave(const double* a, int n) {
  // numpy pairwise sum of n <= 128 contiguous doubles, evaluated by one wave; result valid in every lane
  const int lane = threadIdx.x & 63;
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r = 0.0;
  if (lane < 8) {
    r = a[lane];
    for (int i = 8; i < n - (n % 8); i += 8) r += a[i + lane];
  }
  const double r0 = __shfl(r, 0, 64), r1 = __shfl(r, 1, 64), r2 = __shfl(r, 2, 64), r3 = __shfl(r, 3, 64);
  const double r4 = __shfl(r, 4, 64), r5 = __shfl(r, 5, 64), r6 = __shfl(r, 6, 64), r7 = __shfl(r, 7, 64);
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (int i = n - (n % 8); i < n; ++i) res += a[i];
  return res;
}

// obs_rms.update's batch moments of one observation column, in numpy's axis-0 order (rows added one after the other — a serial
// float64 recurrence).  The LDS reads are issued sixteen at a time and only the additions stay on the dependent chain; left
// to itself the compiler pairs every read with its add and exposes the LDS latency 2 N times.
__device__ __forceinline__ void column_moments(const double* col, int stride, int N, double& bm, double& bv) {
  // two register batches in flight: the reads of batch k + 1 are issued before the additions of batch k
  auto load16 = [&](int r0, double (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = col[(r0 + k < N ? r0 + k : N - 1) * stride];
  };
  double va[16], vb[16];
  double sum = 0.0;
  load16(0, va);
  for (int r0 = 0; r0 < N; r0 += 32) {
    load16(r0 + 16 < N ? r0 + 16 : 0, vb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + k < N) sum += va[k];
    load16(r0 + 32 < N ? r0 + 32 : 0, va);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + 16 + k < N) sum += vb[k];
  }
  bm = sum / (double)N;
  double sq = 0.0;
  // (va holds rows 0..15 again: the last prefetch of the loop above wrapped around)
  for (int r0 = 0; r0 < N; r0 += 32) {
    load16(r0 + 16 < N ? r0 + 16 : 0, vb);
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double d = va[k] - bm; va[k] = d * d; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + k < N) sq += va[k];
    load16(r0 + 32 < N ? r0 + 32 : 0, va);
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double d = vb[k] - bm; vb[k] = d * d; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + 16 + k < N) sq += vb[k];
  }
  bv = sq / (double)N;
}

// the same moments for a column stored CONTIGUOUSLY (persistent rollout: the exchanged observations are transposed into LDS), so
// every read is a ds_read with an immediate offset from one running base; the column is padded with >= 32 readable entries.
__device__ __forceinline__ void column_moments_contig(const double* col, int N, double& bm, double& bv) {
  // One dependent float64 add per row is the floor (numpy's axis-0 order is a single chain); everything else stays off that
  // chain: rows are fetched 16 at a time one batch ahead, and only the last, partial block of 32 carries the `row < N` selects
  // (a select per row on the chain costs more than the add itself).
  double va[16], vb[16];
  double sum = 0.0;
  const int NF = N & ~31;              // rows in full blocks of 32
#pragma unroll
  for (int k = 0; k < 16; ++k) va[k] = col[k];
  int r0 = 0;
  for (; r0 < NF; r0 += 32) {
#pragma unroll
    for (int k = 0; k < 16; ++k) vb[k] = col[r0 + 16 + k];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += va[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) va[k] = col[r0 + 32 + k];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += vb[k];
  }
  if (r0 < N) {
#pragma unroll
    for (int k = 0; k < 16; ++k) vb[k] = col[r0 + 16 + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + k < N) sum += va[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + 16 + k < N) sum += vb[k];
  }
  bm = sum / (double)N;
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) va[k] = col[k];
  for (r0 = 0; r0 < NF; r0 += 32) {
#pragma unroll
    for (int k = 0; k < 16; ++k) vb[k] = col[r0 + 16 + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double d = va[k] - bm; va[k] = d * d; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) sq += va[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) va[k] = col[r0 + 32 + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double d = vb[k] - bm; vb[k] = d * d; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) sq += vb[k];
  }
  if (r0 < N) {
#pragma unroll
    for (int k = 0; k < 16; ++k) vb[k] = col[r0 + 16 + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double d = va[k] - bm; va[k] = d * d; }
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double d = vb[k] - bm; vb[k] = d * d; }
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + k < N) sq += va[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) if (r0 + 16 + k < N) sq += vb[k];
  }
  bv = sq / (double)N;
}

__global__ void __launch_bounds__(1024) norm_step_small_kernel(NormStepArgs a) {
  __shared__ double chunk[NORM_CHUNK];
  __shared__ double vec[2][128], dev2[2][128];
  __shared__ double mean_s[MAX_OBS], var_s[MAX_OBS];
  __shared__ double dens[2];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6;
  const int N = a.N, O = a.O;
  const icrl_norm_t& nm = a.nm;
  const bool has_cost = a.raw_cost != nullptr;
  // ---- one round of loads
  for (int i = tid; i < N * O; i += nt) chunk[i] = a.raw_obs[i];
  if (tid < N) {
    double r = nm.ret[tid], c = has_cost ? nm.cost_ret[tid] : 0.0;
    if (nm.training) {
      r = r * nm.reward_gamma + a.raw_rew[tid];
      if (has_cost) c = c * nm.cost_gamma + (double)a.raw_cost[tid];
    }
    vec[0][tid] = r; vec[1][tid] = c;
  }
  double o_mean = 0.0, o_var = 1.0, o_cnt = 0.0;
  if (tid < O) { o_mean = nm.obs_mean[tid]; o_var = nm.obs_var[tid]; o_cnt = nm.obs_count[0]; }
  __syncthreads();
  if (nm.training) {
    if (tid < O) {           // obs_rms.update: rows added in order (numpy's axis-0 reduction)
      double bm, bv;
      column_moments(chunk + tid, O, N, bm, bv);
      chan_merge(o_mean, o_var, o_cnt, bm, bv, (double)N);
      nm.obs_mean[tid] = o_mean; nm.obs_var[tid] = o_var;
      if (tid == 0) nm.obs_count[0] = (double)N + o_cnt;
    }
    if (wv == 15 || (wv == 14 && has_cost)) {     // ret_rms / cost_rms: numpy pairwise order
      const int v = wv == 15 ? 0 : 1;
      double* st = v == 0 ? nm.ret_stats : nm.cost_stats;
      const double bm = np_leaf_sum_wave(vec[v], N) / (double)N;
      for (int i = lane; i < N; i += 64) { const double d = vec[v][i] - bm; dev2[v][i] = d * d; }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      const double bv = np_leaf_sum_wave(dev2[v], N) / (double)N;
      double m = st[0], var = st[1];
      const double cnt = st[2];
      chan_merge(m, var, cnt, bm, bv, (double)N);
      if (lane == 0) { st[0] = m; st[1] = var; st[2] = (double)N + cnt; }
      if (lane == 0) dens[v] = sqrt(var + nm.epsilon);
    }
  } else {
    if (tid == 0) { dens[0] = sqrt(nm.ret_stats[1] + nm.epsilon); dens[1] = sqrt(nm.cost_stats[1] + nm.epsilon); }
  }
  if (nm.training && !has_cost && tid == 0) dens[1] = sqrt(nm.cost_stats[1] + nm.epsilon);
  if (tid < O) { mean_s[tid] = o_mean; var_s[tid] = o_var; }
  __syncthreads();
  // ---- normalise + clip from LDS
  for (int idx = tid; idx < N * O; idx += nt) {
    const int j = idx % O;
    double o = chunk[idx];
    if (nm.norm_obs) o = fmin(fmax((o - mean_s[j]) / sqrt(var_s[j] + nm.epsilon), -nm.clip_obs), nm.clip_obs);
    if (a.obs_out) a.obs_out[idx] = o;
    if (a.obs_f32) a.obs_f32[idx] = (float)o;
  }
  if (tid < N) {
    const int n = tid;
    double r = a.raw_rew[n];
    if (nm.norm_reward) r = fmin(fmax(r / dens[0], -nm.clip_reward), nm.clip_reward);
    if (a.rew_out) a.rew_out[n] = r;
    if (a.rew_f32) a.rew_f32[n] = (float)r;
    const int d = a.dones[n];
    if (nm.training || d) nm.ret[n] = d ? 0.0 : vec[0][n];
    if (has_cost) {
      double c = (double)a.raw_cost[n];
      if (nm.norm_cost) c = fmin(fmax(c / dens[1], -nm.clip_cost), nm.clip_cost);
      if (a.cost_out) a.cost_out[n] = c;
      if (a.cost_f32) a.cost_f32[n] = (float)c;
      if (nm.training || d) nm.cost_ret[n] = d ? 0.0 : vec[1][n];
    }
    if (a.last_dones) a.last_dones[n] = (uint8_t)d;
  }
}

// =================================================================================================================
// persistent rollout: ALL T steps of collect_rollouts in one launch (N <= 128 envs, N * obs <= NORM_CHUNK)
//
// One workgroup per env runs kernel A's work for its env; the normaliser needs every env's raw observation / reward / cost of
// the step, so the workgroups exchange them through global memory and meet at ONE device-wide barrier per step.  After
// the barrier EVERY workgroup evaluates kernel B's statistics update redundantly — same inputs, same (numpy) summation order,
// hence bit-identical replicas of obs_rms / ret_rms / cost_rms / ret / cost_ret in each workgroup's registers and LDS — and
// normalises only its own env.  The exchange arrays are double-buffered by step parity, which is what makes a single barrier
// per step sufficient (a workgroup can be at most one phase ahead of the slowest one).  Policy / cost-net weights, the
// dynamics matrix and the replicated statistics are loaded once for the whole rollout.
// =================================================================================================================
// diagnostic: cycles per phase of workgroup 0 of the last persistent rollout (policy+env | barrier | statistics), steps
constexpr int NORM_MAX_N_TRACE = 1024;
__device__ unsigned long long g_rollout_prof[8];
__device__ unsigned long long g_rollout_prof_wide[16];
__device__ unsigned long long g_wide_trace[NORM_MAX_N_TRACE * 4];   // per workgroup, one step: s_memrealtime at phase-A end / gather end / publish / statistics read     // rollout_wide_kernel: [0..7] wave 0, [8..15] wave 1 (owner) of the profiled workgroup
__device__ __forceinline__ unsigned long long prof_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

struct PersistArgs {
  ActStepArgs act;
  icrl_norm_t nm;
  int T;
  double* xch_obs;     // [2][N][O]
  double* xch_rew;     // [2][N]
  float* xch_cost;     // [2][N]
  unsigned* xch_done;  // [2][N]
  unsigned* counter;   // [N] barrier flags, zeroed before the launch
  unsigned long long* xg;   // granule mode: [2][N][2 obs + 4] self-validating 8-byte words {step tag | 32 payload bits}, zeroed
  unsigned g_magic;         // floor(2^32 / (2 obs + 4)) for the index split
  int prof;            // diagnostic phase timers (do_gae bit 2)
};

// Exchange between the workgroups of the persistent rollout.  Every exchanged word is written and read with agent-scope
// relaxed atomics (write-through stores / cache-bypassing loads: cdna_hip_programming.md Guideline 16), so no bulk cache
// write-back / invalidate fence is needed — those cost ~4 us per step on this part.  Ordering: the publisher drains its
// stores (s_waitcnt vmcnt(0)) before it raises its flag; consumers read data only after they have seen the flag.
__device__ __forceinline__ void xstore(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xstore(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xstore(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double xload(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float xload(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned xload(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(unsigned long long* p, unsigned tag, unsigned payload) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long gload(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// device-wide barrier over the N (<= 128) workgroups: workgroup n raises its own flag word to `value` (no read-modify-write
// contention on one address), wave 0 polls all N flags with one or two loads per lane.  Flags only grow: no reset.
__device__ __forceinline__ void grid_barrier(unsigned* flags, int N, int n, unsigned value, int& spin_limit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's exchange stores have been performed
  __syncthreads();                                      // ... and everybody else's in the workgroup
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane == 0) xstore(flags + n, value);
    int spins = 0;
    bool ok = false;
    while (spins < spin_limit && !ok) {
      const unsigned v0 = lane < N ? xload(flags + lane) : value;
      const unsigned v1 = lane + 64 < N ? xload(flags + lane + 64) : value;
      ok = __all(v0 >= value && v1 >= value);
      ++spins;
    }
    if (!ok) spin_limit = 1;      // caller reports through icrl_agent_t.status
  }
  __syncthreads();
}

constexpr int GRAN_MAX = 12;    // 8-byte words of exchange area per thread in record mode (N (2 obs + 4) <= 256 * GRAN_MAX) ...
constexpr int REC_MAX = GRAN_MAX / 2;   // ... = 16-byte records a thread polls per step

// GRAN: every exchanged float64 travels in its own 16-byte record with the step number at both ends (the datum is the flag:
// cdna_hip_programming.md Guideline 16; a torn access shows an old tag in one half).  Consumers poll the records themselves, so a
// step costs ONE trip through the memory system after the slowest producer instead of three (drain stores -> raise flag -> see
// flag -> fetch data).  (Until the end of round 3: one 8-byte {tag, 32 bits} granule per half-word — twice the memory
// instructions, and the all-to-all read of 64 workgroups ran into the chip's rate of uncached loads.)
// dynamic LDS of the persistent rollout: the transposed observation block [obs][N + 64] and the dynamics matrix [obs][act], sized
// for the launch's shapes (HC x 64: 19 KB instead of the 114 KB of the largest admissible shape, so that several workgroups —
// several runs of a batched launch — share a CU)
static inline size_t persist_dyn_lds(int N, int O, int A) { return ((size_t)O * (size_t)(N + 64) + (size_t)O * (size_t)A) * sizeof(double); }

// PROF: the diagnostic phase timers (do_gae bit 2) as a compile-time variant: as a run-time flag their seven 64-bit accumulators sat in scalar registers
// across the step loop of every launch (ppo_train_halves.hip: 2.5 % there)
template <int OCT, int CIT, bool GRAN, bool PROF = false>
__device__ __forceinline__ void rollout_persistent_body(const PersistArgs& p) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  __shared__ ActShared sh;
  double* const chunk = dyn_lds;                          // raw observations of the step, TRANSPOSED: [obs][NP], NP = N + 64
  double* const Bl = dyn_lds + p.act.pl.O * (p.act.env.n_envs + 64);
  __shared__ double vec[2][128], dev2[2][128], ret_s[128], cret_s[128], rawr_s[128];
  __shared__ float rawc_s[128];
  __shared__ double dens[2];
  __shared__ float noise_s[MAX_ACT], alow_s[MAX_ACT], ahigh_s[MAX_ACT];     // action box: read every step, kept out of global memory
  __shared__ int done_s[128];
  __shared__ int last_done_s;
  const ActStepArgs& a = p.act;
  // private copies of the structs whose pointers the step loop goes through, every pointer marked as a global-memory pointer
  // (common.h: as_global — in a batched launch the block comes from LDS / memory and the accesses would be flat_* otherwise)
  icrl_norm_t nm = p.nm; globalize(nm);
  icrl_buffer_t buf = a.buf; globalize(buf);
  icrl_agent_t ag = a.ag; globalize(ag);
  icrl_costnet_t cnet = a.cn; globalize(cnet);
  unsigned long long* const xg_all = as_global(p.xg);
  const float* const noise_g = as_global(a.noise);
  WaveRegs<OCT, CIT> R;                // one image: policy weights in waves 0..2, cost-net weights in wave 3
  WaveRegs<OCT, CIT>& C = R;
  load_pol_regs<OCT>(a.pl, a.PT, R);
  if (threadIdx.x >= 192 && a.has_cn) load_cn_regs<CIT>(a.cn, a.cl, C);
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = a.pl.O, A = a.pl.A, N = a.env.n_envs, T = p.T;
  const int AS = buf.act_store;
  const int NA = a.pl.discrete ? 1 : A;       // noise values per env step
  const int NP = N + 64;                      // padded column length of the transposed observation block
  const int G = 2 * O + 4;                    // 8-byte words per env and step of the exchange area
  // Exchange records (GRAN): 16 bytes {tag, lo, hi, tag} — a float64 with this step's tag at both ends (a torn 16-byte access shows an
  // old tag in one half).  Per env: obs_dim observation records, one reward record (the done flag in the top bit of its second tag), one
  // cost record = R16 = obs_dim + 2 records = the same G x 8 bytes as one 8-byte {tag, word} granule per 32-bit word, but HALF the
  // memory instructions: 64 workgroups x 2560 granule loads per step ran into the chip's rate of uncached loads (~62 G/s).
  const int R16 = O + 2;
  typedef unsigned int rec_u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xg_all, 0, 2 * N * G * 8, 0x00020000);
  auto rstore = [&](int byte_off, rec_u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, xrs, byte_off, 0, 16); };     // sc1
  auto rload = [&](int byte_off) -> rec_u4 { return __builtin_amdgcn_raw_buffer_load_b128(xrs, byte_off, 0, 16); };
  for (int i = tid; i < O * NP; i += 256) chunk[i] = 0.0;
  const bool has_cost = a.has_cn != 0;
  const uint32_t e_key = a.env.key[n];
  uint32_t e_ctr = a.env.step_count[n];
  int e_tep = a.env.t_ep[n];
  icrl_env_t env = a.env; globalize(env);
  for (int i = tid; i < O * a.env.act_dim; i += 256) Bl[i] = a.env.B[i];
  env.B = Bl;
  const bool has_box = a.alow != nullptr && a.ahigh != nullptr;
  if (tid < MAX_ACT) { alow_s[tid] = (has_box && tid < A) ? a.alow[tid] : 0.f; ahigh_s[tid] = (has_box && tid < A) ? a.ahigh[tid] : 0.f; }
  for (int i = tid; i < MAX_OBS; i += 256) {
    sh.x[i] = i < O ? (float)ag.last_obs[(size_t)n * O + i] : 0.f;
    if (i < O) sh.s_old[i] = a.env.s[(size_t)n * O + i];
  }
  if (tid < N) { ret_s[tid] = nm.ret[tid]; cret_s[tid] = nm.cost_ret[tid]; }
  if (tid == 0) last_done_s = ag.last_dones[n];
  // replicated running statistics: observation columns in threads < O, ret_rms in wave 3, cost_rms in wave 2
  double o_mean = 0.0, o_var = 1.0, o_cnt = 0.0, o_last = 0.0;
  if (tid < O) { o_mean = nm.obs_mean[tid]; o_var = nm.obs_var[tid]; o_cnt = nm.obs_count[0]; o_last = ag.last_obs[(size_t)n * O + tid]; }
  double st_m = 0.0, st_v = 1.0, st_c = 0.0;
  if (w == 3) { st_m = nm.ret_stats[0]; st_v = nm.ret_stats[1]; st_c = nm.ret_stats[2]; }
  if (w == 2) { st_m = nm.cost_stats[0]; st_v = nm.cost_stats[1]; st_c = nm.cost_stats[2]; }
  float noise_reg = (tid < NA) ? noise_g[(size_t)n * NA + tid] : 0.f;
  double fin_rew = 0.0; float fin_cost = 0.f; int fin_done = 0;
  unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, pc5 = 0, pc_rounds = 0, tl = PROF ? prof_now() : 0ull;
  int spin_limit = 1 << 22;
  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const size_t tn = (size_t)t * N + n;
    const unsigned gtag = (unsigned)(t + 1);
    if (tid < NA) {
      noise_s[tid] = noise_reg;
      if (t + 1 < T) noise_reg = noise_g[((size_t)(t + 1) * N + n) * NA + tid];     // lands during this step
    }
    __syncthreads();
    // ---------------- phase A: kernel A's work for env n ----------------
    policy_forward_block<OCT>(a.pl, R, sh, noise_s, 0, has_box ? alow_s : nullptr, has_box ? ahigh_s : nullptr);
    __syncthreads();
    if (w == 0) {
      double rew; int done;
      env_step_wave(env, n, sh.s_old, sh.act_clip, e_key, e_ctr, e_tep, sh.s_new, rew, done);
      float* nob = buf.new_orig_observations + tn * O;
      if (GRAN) {
        const int rec0 = ((par * N + n) * R16) * 16;       // byte offset of this env's records of this parity
        for (int i = lane; i < O; i += WAVE) {
          const double v = sh.s_new[i];
          nob[i] = (float)v;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
          rstore(rec0 + 16 * i, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag});
        }
        if (lane == 0) {       // reward record: the done flag rides in the top bit of its second tag
          const unsigned long long bits = (unsigned long long)__double_as_longlong(rew);
          rstore(rec0 + 16 * O, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag | (done ? 0x80000000u : 0u)});
        }
      } else {
        double* xo = as_global(p.xch_obs) + ((size_t)par * N + n) * O;
        for (int i = lane; i < O; i += WAVE) { const double v = sh.s_new[i]; nob[i] = (float)v; xstore(xo + i, v); }
        if (lane == 0) { xstore(as_global(p.xch_rew) + par * N + n, rew); xstore(as_global(p.xch_done) + par * N + n, (unsigned)done); }
      }
    } else if (w == 3) {
      float cost = 0.f;
      if (a.has_cn) cost = cost_forward_wave<CIT>(cnet, a.cl, C, sh.s_old, sh.act_clip, sh.cx, sh.ch);
      if (lane == 0) {
        if (GRAN) rstore(((par * N + n) * R16 + O + 1) * 16, rec_u4{gtag, __float_as_uint(cost), 0u, gtag});
        else xstore(as_global(p.xch_cost) + par * N + n, cost);
        buf.orig_costs[tn] = cost;
      }
    } else if (w == 2) {
      float* ob = buf.observations + tn * O;
      float* oob = buf.orig_observations + tn * O;
      for (int i = lane; i < O; i += WAVE) { ob[i] = sh.x[i]; oob[i] = (float)sh.s_old[i]; }
      if (lane < AS) buf.actions[tn * AS + lane] = sh.act_raw[lane];
      if (lane < A && !a.pl.discrete) ag.act_clipped[(size_t)n * A + lane] = sh.act_clip[lane];
      if (lane == 0) {
        buf.dones[tn] = (float)last_done_s;
        buf.reward_values[tn] = sh.scal[0];
        buf.cost_values[tn] = sh.scal[1];
        buf.log_probs[tn] = sh.scal[2];
        ag.last_v_r[n] = sh.scal[0];
        ag.last_v_c[n] = sh.scal[1];
      }
    }
    if (PROF) { const unsigned long long tn_ = prof_now(); pc0 += tn_ - tl; tl = tn_; }
    if (PROF && t == T / 2 && lane == 0) g_wide_trace[4 * n + w] = __builtin_amdgcn_s_memrealtime();   // per wave: end of its phase-A part
    if (GRAN) {
      // poll this thread's records of ALL envs until every one carries this step's tag at both ends, then scatter the payloads
      const int rbase = par * N * R16 * 16;
      const int total = N * R16;
      rec_u4 g[REC_MAX];
#pragma unroll
      for (int k = 0; k < REC_MAX; ++k) g[k] = (k * 256 + tid < total) ? rload(rbase + (k * 256 + tid) * 16) : rec_u4{gtag, 0u, 0u, gtag};
      bool ok = false;
      int rounds = 0;
      for (int spins = 0; spins < spin_limit && !ok; ++spins) {
        ok = true;
#pragma unroll
        for (int k = 0; k < REC_MAX; ++k)
          if (g[k][0] != gtag || (g[k][3] & 0x7fffffffu) != gtag) { g[k] = rload(rbase + (k * 256 + tid) * 16); ok = false; }
        ++rounds;
      }
      if (PROF && tid == 0) { pc_rounds += (unsigned long long)rounds; }
      if (!ok) spin_limit = 1;      // a peer never showed up (a workgroup was not resident): stop waiting ~2 s per step; reported below
#pragma unroll
      for (int k = 0; k < REC_MAX; ++k) {
        const int idx = k * 256 + tid;
        if (idx < total) {
          unsigned rr = __umulhi((unsigned)idx, p.g_magic);
          int slot = idx - (int)rr * R16;
          if (slot >= R16) { slot -= R16; ++rr; }
          const double v = __longlong_as_double((long long)(((unsigned long long)g[k][2] << 32) | (unsigned long long)g[k][1]));
          if (slot < O) chunk[slot * NP + (int)rr] = v;
          else if (slot == O) { rawr_s[rr] = v; done_s[rr] = (int)(g[k][3] >> 31); }
          else rawc_s[rr] = has_cost ? __uint_as_float(g[k][1]) : 0.f;
        }
      }
      __syncthreads();
    } else {
      grid_barrier(as_global(p.counter), N, n, (unsigned)(t + 1), spin_limit);
    }
    if (PROF) { const unsigned long long tn_ = prof_now(); pc1 += tn_ - tl; tl = tn_; }
    // ---------------- phase B: kernel B's statistics, replicated; normalise own env ----------------
    {
      if (!GRAN) {
        const double* xo = as_global(p.xch_obs) + (size_t)par * N * O;
        for (int i = tid; i < N * O; i += 256) { const int rr = i / O, j = i - rr * O; chunk[j * NP + rr] = xload(xo + i); }
      }
      if (tid < N) {
        if (!GRAN) {
          rawr_s[tid] = xload(as_global(p.xch_rew) + par * N + tid);
          rawc_s[tid] = has_cost ? xload(as_global(p.xch_cost) + par * N + tid) : 0.f;
          done_s[tid] = (int)xload(as_global(p.xch_done) + par * N + tid);
        }
        const double rr = rawr_s[tid];
        const float rc = rawc_s[tid];
        double r = ret_s[tid], c = has_cost ? cret_s[tid] : 0.0;
        if (nm.training) {
          r = r * nm.reward_gamma + rr;
          if (has_cost) c = c * nm.cost_gamma + (double)rc;
        }
        vec[0][tid] = r; vec[1][tid] = c;
      }
    }
    __syncthreads();
    if (PROF) { const unsigned long long tn_ = prof_now(); pc3 += tn_ - tl; tl = tn_; }
    if (nm.training) {
      if (tid < O) {           // obs_rms.update: rows added in order (numpy's axis-0 reduction)
        double bm, bv;
        column_moments_contig(chunk + tid * NP, N, bm, bv);
        chan_merge(o_mean, o_var, o_cnt, bm, bv, (double)N);
        o_cnt = (double)N + o_cnt;
      }
      if (w == 3 || (w == 2 && has_cost)) {     // ret_rms / cost_rms: numpy pairwise order
        const int v = w == 3 ? 0 : 1;
        const double bm = np_leaf_sum_wave(vec[v], N) / (double)N;
        for (int i = lane; i < N; i += 64) { const double d = vec[v][i] - bm; dev2[v][i] = d * d; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const double bv = np_leaf_sum_wave(dev2[v], N) / (double)N;
        chan_merge(st_m, st_v, st_c, bm, bv, (double)N);
        st_c = (double)N + st_c;
      }
    }
    if (lane == 0 && w == 3) dens[0] = sqrt(st_v + nm.epsilon);
    if (lane == 0 && w == 2) dens[1] = sqrt(st_v + nm.epsilon);
    if (PROF) { const unsigned long long tn_ = prof_now(); pc4 += tn_ - tl; tl = tn_; }
    if (tid < O) {             // own env: normalise + clip, next policy input
      double o = chunk[tid * NP + n];
      if (nm.norm_obs) o = fmin(fmax((o - o_mean) / sqrt(o_var + nm.epsilon), -nm.clip_obs), nm.clip_obs);
      o_last = o;
      sh.x[tid] = (float)o;
      buf.new_observations[tn * O + tid] = (float)o;
      sh.s_old[tid] = sh.s_new[tid];
    }
    __syncthreads();
    if (PROF) { const unsigned long long tn_ = prof_now(); pc5 += tn_ - tl; tl = tn_; }
    if (tid < N) {
      const int d = done_s[tid];
      if (nm.training || d) { ret_s[tid] = d ? 0.0 : vec[0][tid]; if (has_cost) cret_s[tid] = d ? 0.0 : vec[1][tid]; }
    }
    if (tid == 0) {
      double r = rawr_s[n];
      fin_rew = r; fin_cost = rawc_s[n]; fin_done = done_s[n];
      if (nm.norm_reward) r = fmin(fmax(r / dens[0], -nm.clip_reward), nm.clip_reward);
      buf.rewards[tn] = (float)r;
      if (has_cost) {
        double c = (double)rawc_s[n];
        if (nm.norm_cost) c = fmin(fmax(c / dens[1], -nm.clip_cost), nm.clip_cost);
        buf.costs[tn] = (float)c;
      }
      last_done_s = done_s[n];
    }
    if (PROF) { const unsigned long long tn_ = prof_now(); pc2 += tn_ - tl; tl = tn_; }
  }
  if (PROF && n == 0 && tid == 0) { g_rollout_prof[0] = pc0; g_rollout_prof[1] = pc1; g_rollout_prof[2] = pc2; g_rollout_prof[3] = (unsigned long long)T; g_rollout_prof[4] = pc3; g_rollout_prof[5] = pc4; g_rollout_prof[6] = pc5; g_rollout_prof[7] = pc_rounds; }
  // a timed-out exchange means stale granules went into the statistics and the buffer: tell the host (it raises)
  if (spin_limit == 1 && ag.status != nullptr && (tid & 63) == 0) atomicOr(ag.status, 1);
  // ---- leave the agent / wrapper state exactly where the per-step path leaves it
  __syncthreads();
  if (tid < O) ag.last_obs[(size_t)n * O + tid] = o_last;
  if (tid == 0) {
    ag.last_dones[n] = (uint8_t)last_done_s;
    ag.raw_rew[n] = fin_rew; ag.dones[n] = (uint8_t)fin_done;
    if (has_cost) ag.raw_cost[n] = fin_cost;
  }
  if (n == 0) {
    if (tid < O) { nm.obs_mean[tid] = o_mean; nm.obs_var[tid] = o_var; }
    if (tid == 0) nm.obs_count[0] = o_cnt;
    if (lane == 0 && w == 3) { nm.ret_stats[0] = st_m; nm.ret_stats[1] = st_v; nm.ret_stats[2] = st_c; }
    if (lane == 0 && w == 2 && has_cost) { nm.cost_stats[0] = st_m; nm.cost_stats[1] = st_v; nm.cost_stats[2] = st_c; }
    if (tid < N) { nm.ret[tid] = ret_s[tid]; if (has_cost) nm.cost_ret[tid] = cret_s[tid]; }
  }
}

// The persistent rollout for a generic-shape policy (hidden layers above 64 units or an `arch` descriptor; the constraint net one the
// register image holds): the same loop, phase A's policy forward = the table-driven forward of generic.hip (gen_mlp_forward +
// gen_policy_head: bit-identical to policy_generic_kernel, i.e. to the per-step launches) in its four-units-per-lane form (generic.h:
// gen_mlp_forward_quads — waves 0..2 = the three slots of the table).
struct GenRolloutArgs {
  PersistArgs p;
  GenNet net;
  const float* P;      // the policy's parameters in their natural layout (log_std of the Gaussian head)
};
template <int OCT, int CIT>
__device__ __forceinline__ void rollout_generic_body(const GenRolloutArgs& ga) {
  constexpr bool GRAN = true;
  const PersistArgs& p = ga.p;
  const GenNet& net = ga.net;
  __shared__ float gact[GEN_MAX_ROW];
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  __shared__ ActShared sh;
  double* const chunk = dyn_lds;                          // raw observations of the step, TRANSPOSED: [obs][NP], NP = N + 64
  double* const Bl = dyn_lds + p.act.pl.O * (p.act.env.n_envs + 64);
  __shared__ double vec[2][128], dev2[2][128], ret_s[128], cret_s[128], rawr_s[128];
  __shared__ float rawc_s[128];
  __shared__ double dens[2];
  __shared__ float noise_s[MAX_ACT], alow_s[MAX_ACT], ahigh_s[MAX_ACT];     // action box: read every step, kept out of global memory
  __shared__ int done_s[128];
  __shared__ int last_done_s;
  const ActStepArgs& a = p.act;
  // private copies of the structs whose pointers the step loop goes through, every pointer marked as a global-memory pointer
  // (common.h: as_global — in a batched launch the block comes from LDS / memory and the accesses would be flat_* otherwise)
  icrl_norm_t nm = p.nm; globalize(nm);
  icrl_buffer_t buf = a.buf; globalize(buf);
  icrl_agent_t ag = a.ag; globalize(ag);
  icrl_costnet_t cnet = a.cn; globalize(cnet);
  unsigned long long* const xg_all = as_global(p.xg);
  const float* const noise_g = as_global(a.noise);
  WaveRegs<OCT, CIT> R;                // one image: policy weights in waves 0..2, cost-net weights in wave 3
  WaveRegs<OCT, CIT>& C = R;
  if ((threadIdx.x >> 6) == 3 && a.has_cn) load_cn_regs<CIT>(a.cn, a.cl, C);
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ctid = tid < 256 ? tid : (1 << 28);      // (the threads beyond the first 256 only take part in the forward)
  const float* const Pn = as_global(ga.P);
  const float* const PTn = as_global(a.PT);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = a.pl.O, A = a.pl.A, N = a.env.n_envs, T = p.T;
  const int AS = buf.act_store;
  const int NA = a.pl.discrete ? 1 : A;       // noise values per env step
  const int NP = N + 64;                      // padded column length of the transposed observation block
  const int G = 2 * O + 4;                    // 8-byte words per env and step of the exchange area
  // Exchange records (GRAN): 16 bytes {tag, lo, hi, tag} — a float64 with this step's tag at both ends (a torn 16-byte access shows an
  // old tag in one half).  Per env: obs_dim observation records, one reward record (the done flag in the top bit of its second tag), one
  // cost record = R16 = obs_dim + 2 records = the same G x 8 bytes as one 8-byte {tag, word} granule per 32-bit word, but HALF the
  // memory instructions: 64 workgroups x 2560 granule loads per step ran into the chip's rate of uncached loads (~62 G/s).
  const int R16 = O + 2;
  typedef unsigned int rec_u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xg_all, 0, 2 * N * G * 8, 0x00020000);
  auto rstore = [&](int byte_off, rec_u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, xrs, byte_off, 0, 16); };     // sc1
  auto rload = [&](int byte_off) -> rec_u4 { return __builtin_amdgcn_raw_buffer_load_b128(xrs, byte_off, 0, 16); };
  for (int i = ctid; i < O * NP; i += 256) chunk[i] = 0.0;
  const bool has_cost = a.has_cn != 0;
  const uint32_t e_key = a.env.key[n];
  uint32_t e_ctr = a.env.step_count[n];
  int e_tep = a.env.t_ep[n];
  icrl_env_t env = a.env; globalize(env);
  for (int i = ctid; i < O * a.env.act_dim; i += 256) Bl[i] = a.env.B[i];
  env.B = Bl;
  const bool has_box = a.alow != nullptr && a.ahigh != nullptr;
  if (tid < MAX_ACT) { alow_s[tid] = (has_box && tid < A) ? a.alow[tid] : 0.f; ahigh_s[tid] = (has_box && tid < A) ? a.ahigh[tid] : 0.f; }
  for (int i = ctid; i < MAX_OBS; i += 256) {
    sh.x[i] = i < O ? (float)ag.last_obs[(size_t)n * O + i] : 0.f;
    if (i < O) sh.s_old[i] = a.env.s[(size_t)n * O + i];
  }
  if (tid < N) { ret_s[tid] = nm.ret[tid]; cret_s[tid] = nm.cost_ret[tid]; }
  if (tid == 0) last_done_s = ag.last_dones[n];
  // replicated running statistics: observation columns in threads < O, ret_rms in wave 3, cost_rms in wave 2
  double o_mean = 0.0, o_var = 1.0, o_cnt = 0.0, o_last = 0.0;
  if (tid < O) { o_mean = nm.obs_mean[tid]; o_var = nm.obs_var[tid]; o_cnt = nm.obs_count[0]; o_last = ag.last_obs[(size_t)n * O + tid]; }
  double st_m = 0.0, st_v = 1.0, st_c = 0.0;
  if (w == 3) { st_m = nm.ret_stats[0]; st_v = nm.ret_stats[1]; st_c = nm.ret_stats[2]; }
  if (w == 2) { st_m = nm.cost_stats[0]; st_v = nm.cost_stats[1]; st_c = nm.cost_stats[2]; }
  float noise_reg = (tid < NA) ? noise_g[(size_t)n * NA + tid] : 0.f;
  double fin_rew = 0.0; float fin_cost = 0.f; int fin_done = 0;
  unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, pc5 = 0, pc_rounds = 0, tl = p.prof ? prof_now() : 0ull;
  int spin_limit = 1 << 22;
  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const size_t tn = (size_t)t * N + n;
    const unsigned gtag = (unsigned)(t + 1);
    if (tid < NA) {
      noise_s[tid] = noise_reg;
      if (t + 1 < T) noise_reg = noise_g[((size_t)(t + 1) * N + n) * NA + tid];     // lands during this step
    }
    __syncthreads();
    // ---------------- phase A: kernel A's work for env n ----------------
    gen_mlp_forward_quads(net, PTn, sh.x, gact, w, lane);
    if (tid == 0) {
      float lp, ent;
      gen_policy_head(net, Pn, gact + net.layer[net.head[0]].act_off, noise_s, 0, has_box ? alow_s : nullptr, has_box ? ahigh_s : nullptr, nullptr,
                      sh.act_raw, sh.act_clip, lp, ent);
      sh.scal[0] = gact[net.layer[net.head[1]].act_off]; sh.scal[1] = gact[net.layer[net.head[2]].act_off]; sh.scal[2] = lp;
    }
    __syncthreads();
    if (w == 0) {
      double rew; int done;
      env_step_wave(env, n, sh.s_old, sh.act_clip, e_key, e_ctr, e_tep, sh.s_new, rew, done);
      float* nob = buf.new_orig_observations + tn * O;
      if (GRAN) {
        const int rec0 = ((par * N + n) * R16) * 16;       // byte offset of this env's records of this parity
        for (int i = lane; i < O; i += WAVE) {
          const double v = sh.s_new[i];
          nob[i] = (float)v;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
          rstore(rec0 + 16 * i, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag});
        }
        if (lane == 0) {       // reward record: the done flag rides in the top bit of its second tag
          const unsigned long long bits = (unsigned long long)__double_as_longlong(rew);
          rstore(rec0 + 16 * O, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag | (done ? 0x80000000u : 0u)});
        }
      } else {
        double* xo = as_global(p.xch_obs) + ((size_t)par * N + n) * O;
        for (int i = lane; i < O; i += WAVE) { const double v = sh.s_new[i]; nob[i] = (float)v; xstore(xo + i, v); }
        if (lane == 0) { xstore(as_global(p.xch_rew) + par * N + n, rew); xstore(as_global(p.xch_done) + par * N + n, (unsigned)done); }
      }
    } else if (w == 3) {
      float cost = 0.f;
      if (a.has_cn) cost = cost_forward_wave<CIT>(cnet, a.cl, C, sh.s_old, sh.act_clip, sh.cx, sh.ch);
      if (lane == 0) {
        if (GRAN) rstore(((par * N + n) * R16 + O + 1) * 16, rec_u4{gtag, __float_as_uint(cost), 0u, gtag});
        else xstore(as_global(p.xch_cost) + par * N + n, cost);
        buf.orig_costs[tn] = cost;
      }
    } else if (w == 2) {
      float* ob = buf.observations + tn * O;
      float* oob = buf.orig_observations + tn * O;
      for (int i = lane; i < O; i += WAVE) { ob[i] = sh.x[i]; oob[i] = (float)sh.s_old[i]; }
      if (lane < AS) buf.actions[tn * AS + lane] = sh.act_raw[lane];
      if (lane < A && !a.pl.discrete) ag.act_clipped[(size_t)n * A + lane] = sh.act_clip[lane];
      if (lane == 0) {
        buf.dones[tn] = (float)last_done_s;
        buf.reward_values[tn] = sh.scal[0];
        buf.cost_values[tn] = sh.scal[1];
        buf.log_probs[tn] = sh.scal[2];
        ag.last_v_r[n] = sh.scal[0];
        ag.last_v_c[n] = sh.scal[1];
      }
    }
    if (p.prof) { const unsigned long long tn_ = prof_now(); pc0 += tn_ - tl; tl = tn_; }
    if (p.prof && t == T / 2 && lane == 0) g_wide_trace[4 * n + w] = __builtin_amdgcn_s_memrealtime();   // per wave: end of its phase-A part
    if (GRAN) {
      // poll this thread's records of ALL envs until every one carries this step's tag at both ends, then scatter the payloads
      const int rbase = par * N * R16 * 16;
      const int total = N * R16;
      rec_u4 g[REC_MAX];
#pragma unroll
      for (int k = 0; k < REC_MAX; ++k) g[k] = (k * 256 + ctid < total) ? rload(rbase + (k * 256 + ctid) * 16) : rec_u4{gtag, 0u, 0u, gtag};
      bool ok = false;
      int rounds = 0;
      for (int spins = 0; spins < spin_limit && !ok; ++spins) {
        ok = true;
#pragma unroll
        for (int k = 0; k < REC_MAX; ++k)
          if (g[k][0] != gtag || (g[k][3] & 0x7fffffffu) != gtag) { g[k] = rload(rbase + (k * 256 + ctid) * 16); ok = false; }
        ++rounds;
      }
      if (p.prof && tid == 0) { pc_rounds += (unsigned long long)rounds; }
      if (!ok) spin_limit = 1;      // a peer never showed up (a workgroup was not resident): stop waiting ~2 s per step; reported below
#pragma unroll
      for (int k = 0; k < REC_MAX; ++k) {
        const int idx = k * 256 + ctid;
        if (idx < total) {
          unsigned rr = __umulhi((unsigned)idx, p.g_magic);
          int slot = idx - (int)rr * R16;
          if (slot >= R16) { slot -= R16; ++rr; }
          const double v = __longlong_as_double((long long)(((unsigned long long)g[k][2] << 32) | (unsigned long long)g[k][1]));
          if (slot < O) chunk[slot * NP + (int)rr] = v;
          else if (slot == O) { rawr_s[rr] = v; done_s[rr] = (int)(g[k][3] >> 31); }
          else rawc_s[rr] = has_cost ? __uint_as_float(g[k][1]) : 0.f;
        }
      }
      __syncthreads();
    } else {
      grid_barrier(as_global(p.counter), N, n, (unsigned)(t + 1), spin_limit);
    }
    if (p.prof) { const unsigned long long tn_ = prof_now(); pc1 += tn_ - tl; tl = tn_; }
    // ---------------- phase B: kernel B's statistics, replicated; normalise own env ----------------
    {
      if (!GRAN) {
        const double* xo = as_global(p.xch_obs) + (size_t)par * N * O;
        for (int i = tid; i < N * O; i += 256) { const int rr = i / O, j = i - rr * O; chunk[j * NP + rr] = xload(xo + i); }
      }
      if (tid < N) {
        if (!GRAN) {
          rawr_s[tid] = xload(as_global(p.xch_rew) + par * N + tid);
          rawc_s[tid] = has_cost ? xload(as_global(p.xch_cost) + par * N + tid) : 0.f;
          done_s[tid] = (int)xload(as_global(p.xch_done) + par * N + tid);
        }
        const double rr = rawr_s[tid];
        const float rc = rawc_s[tid];
        double r = ret_s[tid], c = has_cost ? cret_s[tid] : 0.0;
        if (nm.training) {
          r = r * nm.reward_gamma + rr;
          if (has_cost) c = c * nm.cost_gamma + (double)rc;
        }
        vec[0][tid] = r; vec[1][tid] = c;
      }
    }
    __syncthreads();
    if (p.prof) { const unsigned long long tn_ = prof_now(); pc3 += tn_ - tl; tl = tn_; }
    if (nm.training) {
      if (tid < O) {           // obs_rms.update: rows added in order (numpy's axis-0 reduction)
        double bm, bv;
        column_moments_contig(chunk + tid * NP, N, bm, bv);
        chan_merge(o_mean, o_var, o_cnt, bm, bv, (double)N);
        o_cnt = (double)N + o_cnt;
      }
      if (w == 3 || (w == 2 && has_cost)) {     // ret_rms / cost_rms: numpy pairwise order
        const int v = w == 3 ? 0 : 1;
        const double bm = np_leaf_sum_wave(vec[v], N) / (double)N;
        for (int i = lane; i < N; i += 64) { const double d = vec[v][i] - bm; dev2[v][i] = d * d; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const double bv = np_leaf_sum_wave(dev2[v], N) / (double)N;
        chan_merge(st_m, st_v, st_c, bm, bv, (double)N);
        st_c = (double)N + st_c;
      }
    }
    if (lane == 0 && w == 3) dens[0] = sqrt(st_v + nm.epsilon);
    if (lane == 0 && w == 2) dens[1] = sqrt(st_v + nm.epsilon);
    if (p.prof) { const unsigned long long tn_ = prof_now(); pc4 += tn_ - tl; tl = tn_; }
    if (tid < O) {             // own env: normalise + clip, next policy input
      double o = chunk[tid * NP + n];
      if (nm.norm_obs) o = fmin(fmax((o - o_mean) / sqrt(o_var + nm.epsilon), -nm.clip_obs), nm.clip_obs);
      o_last = o;
      sh.x[tid] = (float)o;
      buf.new_observations[tn * O + tid] = (float)o;
      sh.s_old[tid] = sh.s_new[tid];
    }
    __syncthreads();
    if (p.prof) { const unsigned long long tn_ = prof_now(); pc5 += tn_ - tl; tl = tn_; }
    if (tid < N) {
      const int d = done_s[tid];
      if (nm.training || d) { ret_s[tid] = d ? 0.0 : vec[0][tid]; if (has_cost) cret_s[tid] = d ? 0.0 : vec[1][tid]; }
    }
    if (tid == 0) {
      double r = rawr_s[n];
      fin_rew = r; fin_cost = rawc_s[n]; fin_done = done_s[n];
      if (nm.norm_reward) r = fmin(fmax(r / dens[0], -nm.clip_reward), nm.clip_reward);
      buf.rewards[tn] = (float)r;
      if (has_cost) {
        double c = (double)rawc_s[n];
        if (nm.norm_cost) c = fmin(fmax(c / dens[1], -nm.clip_cost), nm.clip_cost);
        buf.costs[tn] = (float)c;
      }
      last_done_s = done_s[n];
    }
    if (p.prof) { const unsigned long long tn_ = prof_now(); pc2 += tn_ - tl; tl = tn_; }
  }
  if (p.prof && n == 0 && tid == 0) { g_rollout_prof[0] = pc0; g_rollout_prof[1] = pc1; g_rollout_prof[2] = pc2; g_rollout_prof[3] = (unsigned long long)T; g_rollout_prof[4] = pc3; g_rollout_prof[5] = pc4; g_rollout_prof[6] = pc5; g_rollout_prof[7] = pc_rounds; }
  // a timed-out exchange means stale granules went into the statistics and the buffer: tell the host (it raises)
  if (spin_limit == 1 && ag.status != nullptr && (tid & 63) == 0) atomicOr(ag.status, 1);
  // ---- leave the agent / wrapper state exactly where the per-step path leaves it
  __syncthreads();
  if (tid < O) ag.last_obs[(size_t)n * O + tid] = o_last;
  if (tid == 0) {
    ag.last_dones[n] = (uint8_t)last_done_s;
    ag.raw_rew[n] = fin_rew; ag.dones[n] = (uint8_t)fin_done;
    if (has_cost) ag.raw_cost[n] = fin_cost;
  }
  if (n == 0) {
    if (tid < O) { nm.obs_mean[tid] = o_mean; nm.obs_var[tid] = o_var; }
    if (tid == 0) nm.obs_count[0] = o_cnt;
    if (lane == 0 && w == 3) { nm.ret_stats[0] = st_m; nm.ret_stats[1] = st_v; nm.ret_stats[2] = st_c; }
    if (lane == 0 && w == 2 && has_cost) { nm.cost_stats[0] = st_m; nm.cost_stats[1] = st_v; nm.cost_stats[2] = st_c; }
    if (tid < N) { nm.ret[tid] = ret_s[tid]; if (has_cost) nm.cost_ret[tid] = cret_s[tid]; }
  }
}

template <int OCT, int CIT>
__global__ void __launch_bounds__(256) rollout_generic_kernel(GenRolloutArgs ga) {
  rollout_generic_body<OCT, CIT>(ga);
}

template <int OCT, int CIT, bool GRAN, bool PROF = false>
__global__ void __launch_bounds__(256) rollout_persistent_kernel(PersistArgs p) {
  rollout_persistent_body<OCT, CIT, GRAN, PROF>(p);
}

// several independent runs in ONE launch: grid (N, n_runs), run = blockIdx.y, argument blocks in device memory.  Workgroups are
// dispatched x-fastest, so a run's N workgroups become resident together and the oldest run of the grid is always complete: runs
// whose workgroups do not fit yet simply start when earlier runs have finished (the exchange waits are bounded by seconds).
// MINW: waves per SIMD the register allocation must leave room for (= workgroups per CU: each workgroup has one wave per SIMD)
template <int OCT, int CIT, bool GRAN, int MINW>
__global__ void __launch_bounds__(256, MINW) rollout_persistent_batch_kernel(const PersistArgs* __restrict__ runs) {
  __shared__ PersistArgs p;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(runs + blockIdx.y);
    unsigned* dst = reinterpret_cast<unsigned*>(&p);
    for (unsigned i = threadIdx.x; i < sizeof(PersistArgs) / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  rollout_persistent_body<OCT, CIT, GRAN>(p);
}

// =================================================================================================================
// persistent rollout for MANY environments (128 < N <= 1024, or N x obs > 4096): BASELINE configs[2..4] per-GPU shards
// =================================================================================================================
// Same contract as rollout_persistent_kernel (ALL T steps in one launch, buffers / normaliser / agent state bit-identical to
// the per-step launches), different statistics phase.  Replicating the whole [N, obs] observation block in every workgroup
// costs N x obs granule reads per workgroup and step (59 k at AntWall x 256); here the float64 statistics are PARTITIONED:
//   workgroup j < obs        owns observation column j: gathers that column from all N envs (2 N granules), runs numpy's
//                            axis-0 reduction for it (one sequential chain per column: the order, hence every bit, is kept),
//                            merges into the running moments and publishes (mean_j, var_j) as 4 granules;
//   workgroup obs, obs + 1   own ret_rms / cost_rms: discounted returns of all envs, numpy's pairwise sums, merged moments ->
//                            the two normalisation denominators as granules;
//   every workgroup          then reads the 4 obs + 4 statistics granules and normalises its own envs.
// Two hops per step instead of one, but 2 N + 4 obs granules per workgroup instead of N (2 obs + 4).  A workgroup serves
// E = ceil(N / grid) envs one after the other (grid <= what is co-resident: one workgroup per CU at AntWall widths).
constexpr int WIDE_E = 4;

struct WideArgs {
  ActStepArgs act;
  icrl_norm_t nm;
  int T, G;                    // steps; grid size (workgroup g serves envs g, g + G, ...)
  int prof;                    // diagnostic phase timers of workgroup `prof - 1` (do_gae bit 2: workgroup 0; bit 3: the last one)
  unsigned long long* xg;      // [2][N][2 obs + 4] env granules {step tag | 32 payload bits}, zeroed before the launch
  unsigned long long* sg;      // [2][4 obs + 4] statistics granules, zeroed before the launch
  unsigned long long* xcc;     // multi-env kernel, packed batched launches: G zeroed words (the workgroups' XCD ids), else NULL
};

template <int OCT, int CIT, bool PROF = false>      // PROF: the phase timers (do_gae bits 2 / 3) as a compile-time variant (rollout_persistent_body)
__global__ void __launch_bounds__(256) rollout_wide_kernel(WideArgs p) {
  __shared__ ActShared sh[WIDE_E];
  __shared__ double Bl[MAX_OBS * MAX_ACT];
  __shared__ double colbuf[WIDE_MAX_N + 64], dev2buf[WIDE_MAX_N], retbuf[WIDE_MAX_N];
  __shared__ double mean_s[MAX_OBS], var_s[MAX_OBS], dens_s[2];
  __shared__ double olast[WIDE_E][MAX_OBS], rew_s[WIDE_E];
  __shared__ float noise_s[2][WIDE_E][MAX_ACT], alow_s[MAX_ACT], ahigh_s[MAX_ACT], cost_s[WIDE_E];
  __shared__ unsigned ctr_s[WIDE_E];
  __shared__ int tep_s[WIDE_E], last_done_s[WIDE_E], done_s[WIDE_E], done_all[WIDE_MAX_N];
  const ActStepArgs& a = p.act;
  const icrl_norm_t& nm = p.nm;
  WaveRegs<OCT, CIT> R;                // one image: policy weights in waves 0..2, cost-net weights in wave 3
  WaveRegs<OCT, CIT>& C = R;
  load_pol_regs<OCT>(a.pl, a.PT, R);
  if (threadIdx.x >= 192 && a.has_cn) load_cn_regs<CIT>(a.cn, a.cl, C);
  const int g = blockIdx.x, G = p.G;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = a.pl.O, A = a.pl.A, N = a.env.n_envs, T = p.T;
  const int AS = a.buf.act_store;
  const int NA = a.pl.discrete ? 1 : A;
  const int GX = 2 * O + 4, GS = 4 * O + 4;
  // Exchange records: 16 bytes {tag, lo, hi, tag} instead of two 8-byte {tag, word} granules per float64 (same bytes, half the memory
  // instructions; the done flag rides in the top bit of the reward record's second tag).  Per env R16 = obs + 2 records (obs columns,
  // reward, cost); statistics: (mean, var) per column, then the ret / cost denominators.
  const int R16 = O + 2;
  typedef unsigned int rec_u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(p.xg, 0, 2 * N * GX * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(p.sg, 0, 2 * GS * 8, 0x00020000);
  auto rstore = [](const __amdgpu_buffer_rsrc_t& rs, int byte_off, rec_u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16); };   // sc1
  auto rload = [](const __amdgpu_buffer_rsrc_t& rs, int byte_off) -> rec_u4 { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); };
  const int E = (N - g + G - 1) / G;                 // envs of this workgroup (>= 1: G <= N)
  const bool has_cost = a.has_cn != 0;
  const bool has_box = a.alow != nullptr && a.ahigh != nullptr;
  icrl_env_t env = a.env;
  for (int i = tid; i < O * a.env.act_dim; i += 256) Bl[i] = a.env.B[i];
  env.B = Bl;
  if (tid < MAX_ACT) { alow_s[tid] = (has_box && tid < A) ? a.alow[tid] : 0.f; ahigh_s[tid] = (has_box && tid < A) ? a.ahigh[tid] : 0.f; }
  for (int e = 0; e < E; ++e) {
    const int n = g + e * G;
    for (int i = tid; i < MAX_OBS; i += 256) {
      sh[e].x[i] = i < O ? (float)a.ag.last_obs[(size_t)n * O + i] : 0.f;
      if (i < O) { sh[e].s_old[i] = a.env.s[(size_t)n * O + i]; olast[e][i] = a.ag.last_obs[(size_t)n * O + i]; }
    }
    if (tid == 0) { ctr_s[e] = a.env.step_count[n]; tep_s[e] = a.env.t_ep[n]; last_done_s[e] = a.ag.last_dones[n]; }
  }
  // owner roles of this workgroup (its wave 1 does the owner work)
  const int own_col = g < O ? g : -1;
  const bool own_ret = g == O, own_cost = has_cost && g == O + 1;
  double o_mean = 0.0, o_var = 1.0, o_cnt = 0.0;         // wave 1: running moments of the owned statistic
  if (own_col >= 0) { o_mean = nm.obs_mean[own_col]; o_var = nm.obs_var[own_col]; o_cnt = nm.obs_count[0]; }
  if (own_ret) { o_mean = nm.ret_stats[0]; o_var = nm.ret_stats[1]; o_cnt = nm.ret_stats[2]; }
  if (own_cost) { o_mean = nm.cost_stats[0]; o_var = nm.cost_stats[1]; o_cnt = nm.cost_stats[2]; }
  if (own_ret || own_cost)
    for (int i = tid; i < N; i += 256) retbuf[i] = own_ret ? nm.ret[i] : nm.cost_ret[i];
  // noise of step 0
  if (tid < WIDE_E * MAX_ACT) {
    const int e = tid / MAX_ACT, k = tid % MAX_ACT;
    if (e < E && k < NA) noise_s[0][e][k] = a.noise[((size_t)(g + e * G)) * NA + k];
  }
  int spin_limit = 1 << 22;
  __syncthreads();
  const bool prof = PROF && p.prof != 0 && g == p.prof - 1;
  unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, tl = prof ? prof_now() : 0ull;
  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const unsigned gtag = (unsigned)(t + 1);
    // prefetch the next step's noise (lands during this step)
    float noise_next = 0.f;
    const int pe = tid / MAX_ACT, pk = tid % MAX_ACT;
    const bool pf = tid < WIDE_E * MAX_ACT && pe < E && pk < NA && t + 1 < T;
    if (pf) noise_next = a.noise[((size_t)(t + 1) * N + g + pe * G) * NA + pk];
    // ---------------- phase A: policy forward, env step, cost, buffer rows for each env of this workgroup ----------------
    for (int e = 0; e < E; ++e) {
      const int n = g + e * G;
      const size_t tn = (size_t)t * N + n;
      policy_forward_block<OCT>(a.pl, R, sh[e], noise_s[par][e], 0, has_box ? alow_s : nullptr, has_box ? ahigh_s : nullptr);
      __syncthreads();
      const int xrec = (par * N + n) * R16 * 16;      // byte offset of the env's records of this parity
      if (w == 0) {
        double rew; int done;
        uint32_t e_ctr = ctr_s[e];
        int e_tep = tep_s[e];
        env_step_wave(env, n, sh[e].s_old, sh[e].act_clip, a.env.key[n], e_ctr, e_tep, sh[e].s_new, rew, done);
        float* nob = a.buf.new_orig_observations + tn * O;
        for (int i = lane; i < O; i += WAVE) {
          const double v = sh[e].s_new[i];
          nob[i] = (float)v;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
          rstore(xrs, xrec + 16 * i, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag});
        }
        if (lane == 0) {
          const unsigned long long bits = (unsigned long long)__double_as_longlong(rew);
          rstore(xrs, xrec + 16 * O, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag | (done ? 0x80000000u : 0u)});
          ctr_s[e] = e_ctr; tep_s[e] = e_tep; rew_s[e] = rew; done_s[e] = done;
        }
      } else if (w == 3) {
        float cost = 0.f;
        if (a.has_cn) cost = cost_forward_wave<CIT>(a.cn, a.cl, C, sh[e].s_old, sh[e].act_clip, sh[e].cx, sh[e].ch);
        if (lane == 0) {
          rstore(xrs, xrec + 16 * (O + 1), rec_u4{gtag, __float_as_uint(cost), 0u, gtag});
          a.buf.orig_costs[tn] = cost;
          cost_s[e] = cost;
        }
      } else if (w == 2) {
        float* ob = a.buf.observations + tn * O;
        float* oob = a.buf.orig_observations + tn * O;
        for (int i = lane; i < O; i += WAVE) { ob[i] = sh[e].x[i]; oob[i] = (float)sh[e].s_old[i]; }
        if (lane < AS) a.buf.actions[tn * AS + lane] = sh[e].act_raw[lane];
        if (lane < A && !a.pl.discrete) a.ag.act_clipped[(size_t)n * A + lane] = sh[e].act_clip[lane];
        if (lane == 0) {
          a.buf.dones[tn] = (float)last_done_s[e];
          a.buf.reward_values[tn] = sh[e].scal[0];
          a.buf.cost_values[tn] = sh[e].scal[1];
          a.buf.log_probs[tn] = sh[e].scal[2];
          a.ag.last_v_r[n] = sh[e].scal[0];
          a.ag.last_v_c[n] = sh[e].scal[1];
        }
      }
    }
    if (prof) { const unsigned long long tn_ = prof_now(); pc0 += tn_ - tl; tl = tn_; }     // policy + env + rows (this wave's part)
    const bool trace = PROF && p.prof != 0 && t == T / 2 && lane == 0;
    if (trace && w == 0) g_wide_trace[4 * g + 0] = __builtin_amdgcn_s_memrealtime();
    // ---------------- phase B1: the owners (wave 1) gather their statistic from all envs and publish it ----------------
    if (w == 1 && (own_col >= 0 || own_ret || own_cost)) {
      const int slot = own_col >= 0 ? own_col : (own_ret ? O : O + 1);      // the record of every env this owner gathers
      const bool wide = !own_cost;                                   // a float64 in the record, or the float32 cost
      const int gbase = par * N * R16 * 16;
      // four blocks of 64 envs per polling round: all their loads are in flight together (one trip through the memory system
      // per round instead of one per block)
      for (int i0 = 0; i0 < N; i0 += 4 * WAVE) {
        rec_u4 g0[4], gd[4];          // gd: the reward record of the env — the cost owner takes the done flag from its tag
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { g0[k] = gd[k] = rec_u4{0u, 0u, 0u, 0u}; ok[k] = i0 + k * WAVE + lane >= N; }
        for (int spins = 0; spins < spin_limit; ++spins) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * WAVE + lane;
            if (!ok[k]) { g0[k] = rload(xrs, gbase + (i * R16 + slot) * 16); gd[k] = own_cost ? rload(xrs, gbase + (i * R16 + O) * 16) : g0[k]; }
          }
          bool all = true;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (!ok[k]) ok[k] = g0[k][0] == gtag && (g0[k][3] & 0x7fffffffu) == gtag && gd[k][0] == gtag && (gd[k][3] & 0x7fffffffu) == gtag;
            all = all && ok[k];
          }
          if (__all(all)) break;
          if (spins + 1 == spin_limit) spin_limit = 1;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * WAVE + lane;
          if (i < N) {
            double v;
            if (wide) v = __longlong_as_double((long long)(((unsigned long long)g0[k][2] << 32) | (unsigned long long)g0[k][1]));
            else v = (double)__uint_as_float(g0[k][1]);
            if (own_col >= 0) colbuf[i] = v;
            else {
              const double ret = retbuf[i] * (own_ret ? nm.reward_gamma : nm.cost_gamma) + v;     // vec_normalize.py:102, 245
              colbuf[i] = ret;
              done_all[i] = (int)(gd[k][3] >> 31);
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if (prof) { const unsigned long long tn_ = prof_now(); pc1 += tn_ - tl; tl = tn_; }   // owner: gather (wait for all envs)
      if (trace) g_wide_trace[4 * g + 1] = __builtin_amdgcn_s_memrealtime();
      double bm, bv;
      if (own_col >= 0) {
        column_moments_contig(colbuf, N, bm, bv);                   // every lane computes the same chain (numpy's axis-0 order)
      } else {
        bm = np_pairwise_sum_wave(colbuf, N) / (double)N;           // numpy's 1-D pairwise order
        for (int i = lane; i < N; i += WAVE) { const double d = colbuf[i] - bm; dev2buf[i] = d * d; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        bv = np_pairwise_sum_wave(dev2buf, N) / (double)N;
      }
      chan_merge(o_mean, o_var, o_cnt, bm, bv, (double)N);
      o_cnt = (double)N + o_cnt;
      const int sbase = par * (GS / 2) * 16;          // statistics records of this parity: (mean, var) per column, then the two denominators
      if (lane == 0) {
        if (own_col >= 0) {
          const unsigned long long mb = (unsigned long long)__double_as_longlong(o_mean), vb = (unsigned long long)__double_as_longlong(o_var);
          rstore(srs, sbase + 32 * own_col, rec_u4{gtag, (unsigned)mb, (unsigned)(mb >> 32), gtag});
          rstore(srs, sbase + 32 * own_col + 16, rec_u4{gtag, (unsigned)vb, (unsigned)(vb >> 32), gtag});
        } else {
          const unsigned long long db = (unsigned long long)__double_as_longlong(sqrt(o_var + nm.epsilon));
          rstore(srs, sbase + (2 * O + (own_ret ? 0 : 1)) * 16, rec_u4{gtag, (unsigned)db, (unsigned)(db >> 32), gtag});
        }
      }
      if (own_col < 0)      // returns of finished episodes restart at 0 (vec_normalize.py:99, 241)
        for (int i = lane; i < N; i += WAVE) retbuf[i] = done_all[i] ? 0.0 : colbuf[i];
      if (prof) { const unsigned long long tn_ = prof_now(); pc2 += tn_ - tl; tl = tn_; }   // owner: moments + merge + publish
      if (trace) g_wide_trace[4 * g + 2] = __builtin_amdgcn_s_memrealtime();
    }
    // ---------------- phase B2: everybody reads the statistics granules, then normalises its own envs ----------------
    {
      const int sbase = par * (GS / 2) * 16;
      const int total = has_cost ? GS / 2 : GS / 2 - 1;              // records: (mean, var) per column, ret denominator, cost denominator
      for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + tid;
        rec_u4 gv = rec_u4{gtag, 0u, 0u, gtag};
        bool ok = i >= total;
        for (int spins = 0; spins < spin_limit && !ok; ++spins) {
          gv = rload(srs, sbase + i * 16);
          ok = gv[0] == gtag && gv[3] == gtag;
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) spin_limit = 1;
        if (i < total) {
          const double v = __longlong_as_double((long long)(((unsigned long long)gv[2] << 32) | (unsigned long long)gv[1]));
          if (i < 2 * O) { if (i & 1) var_s[i >> 1] = v; else mean_s[i >> 1] = v; }
          else dens_s[i - 2 * O] = v;
        }
      }
    }
    __syncthreads();
    if (prof) { const unsigned long long tn_ = prof_now(); pc3 += tn_ - tl; tl = tn_; }     // statistics granules arrived
    if (trace && w == 0) g_wide_trace[4 * g + 3] = __builtin_amdgcn_s_memrealtime();
    for (int e = 0; e < E; ++e) {
      const int n = g + e * G;
      const size_t tn = (size_t)t * N + n;
      if (tid < O) {
        double o = sh[e].s_new[tid];
        if (nm.norm_obs) o = fmin(fmax((o - mean_s[tid]) / sqrt(var_s[tid] + nm.epsilon), -nm.clip_obs), nm.clip_obs);
        olast[e][tid] = o;
        sh[e].x[tid] = (float)o;
        a.buf.new_observations[tn * O + tid] = (float)o;
        sh[e].s_old[tid] = sh[e].s_new[tid];
      }
      if (tid == 64) {
        double r = rew_s[e];
        if (nm.norm_reward) r = fmin(fmax(r / dens_s[0], -nm.clip_reward), nm.clip_reward);
        a.buf.rewards[tn] = (float)r;
        if (has_cost) {
          double c = (double)cost_s[e];
          if (nm.norm_cost) c = fmin(fmax(c / dens_s[1], -nm.clip_cost), nm.clip_cost);
          a.buf.costs[tn] = (float)c;
        }
        last_done_s[e] = done_s[e];
      }
    }
    if (pf) noise_s[par ^ 1][pe][pk] = noise_next;
    __syncthreads();
    if (prof) { const unsigned long long tn_ = prof_now(); pc4 += tn_ - tl; tl = tn_; }     // normalise + rows
  }
  if (prof && (tid == 0 || tid == 64)) {      // wave 0: policy wave's view; wave 1: the owner's
    unsigned long long* o = g_rollout_prof_wide + (tid == 64 ? 8 : 0);
    o[0] = pc0; o[1] = pc1; o[2] = pc2; o[3] = pc3; o[4] = pc4; o[5] = (unsigned long long)T;
  }
  if (spin_limit == 1 && a.ag.status != nullptr && (tid & 63) == 0) atomicOr(a.ag.status, 1);
  // ---- leave the agent / wrapper / normaliser state exactly where the per-step path leaves it
  for (int e = 0; e < E; ++e) {
    const int n = g + e * G;
    if (tid < O) a.ag.last_obs[(size_t)n * O + tid] = olast[e][tid];
    if (tid == 0) {
      a.ag.last_dones[n] = (uint8_t)last_done_s[e];
      a.ag.raw_rew[n] = rew_s[e]; a.ag.dones[n] = (uint8_t)done_s[e];
      if (has_cost) a.ag.raw_cost[n] = cost_s[e];
    }
  }
  if (w == 1 && lane == 0) {
    if (own_col >= 0) { nm.obs_mean[own_col] = o_mean; nm.obs_var[own_col] = o_var; if (own_col == 0) nm.obs_count[0] = o_cnt; }
    if (own_ret) { nm.ret_stats[0] = o_mean; nm.ret_stats[1] = o_var; nm.ret_stats[2] = o_cnt; }
    if (own_cost) { nm.cost_stats[0] = o_mean; nm.cost_stats[1] = o_var; nm.cost_stats[2] = o_cnt; }
  }
  if (w == 1 && (own_ret || own_cost))
    for (int i = lane; i < N; i += WAVE) { if (own_ret) nm.ret[i] = retbuf[i]; else nm.cost_ret[i] = retbuf[i]; }
}

// =================================================================================================================
// persistent rollout with SEVERAL environments per workgroup, evaluated INTERLEAVED (throughput form)
// =================================================================================================================
// Same contract and bit-identical results as rollout_persistent_kernel / rollout_wide_kernel (every per-env operation is the
// same arithmetic: the k-ascending fmaf chains of the other kernels, evaluated here as fp32 MFMA tiles — TileRegs below —, the same
// tanh, the same float64 env step, numpy-ordered statistics), different use of the machine.  Those kernels give one environment a
// workgroup (or walk a workgroup's environments one after the other): an env step is then a 7-8 k cycle chain of dependent
// latencies.  Here workgroup g of a run serves E environments g, g + G, ... and runs every layer for all of them at once (16 units
// x 16 envs per MFMA, the weights register-resident as A operands, the activations read from LDS as B operands).  E environments
// cost little more than one, so a run needs G = N / E workgroups instead of N, all runs of a
// batched launch (icrl_rollout_collect_batch, run = blockIdx.y) are resident together, and the exchange shrinks with G:
//   phase A   layers 1, 2, heads for E envs (waves pi | vf | cvf); env steps spread over waves 0..2, cost net on wave 3;
//             raw obs / reward / cost / done published as self-validating 16-byte records (as in the other two kernels)
//   phase B1  statistic s (observation column, ret_rms, cost_rms) is owned by wave s / G of workgroup s % G: up to four
//             owners per workgroup work side by side (gather 2 N granules, numpy-ordered moments, merge, publish)
//   phase B2  every workgroup reads the 4 obs + 4 statistics granules and normalises its E envs
constexpr int MULTI_OP = 128;     // padded per-env row of the LDS state arrays (>= MAX_OBS)
// rows that feed the MFMA B operand (lane (r, q) reads element 4 ks + q of row r): a row stride of 4 mod 32 floats spreads the 64
// lanes of a read over all banks (two lanes per bank: the minimum); strides of 64 / 128 would put 16 lanes on one bank
constexpr int MULTI_XS = MULTI_OP + 4;
constexpr int MULTI_HS = MAX_H + 4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// per-wave constants read back as 16-byte rows where they are used (they would cost 84 registers per lane next to the weights):
// by unit j: b1 | b2 | output weight (value / cost-value / cost net);  by action a (wave 0): bias | exp(log_std) | log(sd) | 2 sd^2;  [CST_BO]: output bias
constexpr int CST_B1 = 0, CST_B2 = 64, CST_WO = 128, CST_BA = 192, CST_SD = 208, CST_LSD = 224, CST_I2V = 240, CST_BO = 256, MULTI_CST = 260;

template <int E, int CIT>
struct MultiShared {
  alignas(16) float x[E][MULTI_XS];          // normalised observation (policy input), pad = 0
  alignas(16) double s_old[E][MULTI_OP];
  alignas(16) double s_new[E][MULTI_OP];
  alignas(16) double olast[E][MULTI_OP];     // normalised observation in float64 (== _last_obs)
  alignas(16) float h[4][E][MULTI_HS];       // per wave (pi | vf | cvf | cost net): the layer output it turns into its next B operand
  alignas(16) float cx[E][16 * CIT + 4];     // cost-net inputs of every env
  alignas(16) float cst[4][MULTI_CST];       // per wave: biases / output weights by unit, Gaussian head constants by action (TileRegs)
  alignas(16) float act_raw[E][MAX_ACT];
  alignas(16) float act_clip[E][MAX_ACT];
  alignas(16) float noise[2][E][MAX_ACT];
  alignas(16) float alow[MAX_ACT];
  alignas(16) float ahigh[MAX_ACT];
  int xcd_local;                             // all workgroups of this run share this XCD (rollout_multi_body)
  float scal[E][4], cost[E];
  double rew[E], mean[MAX_OBS], var[MAX_OBS], dens[2];
  unsigned ctr[E], key[E];
  int tep[E], last_done[E], done[E];
};

// Register image of one wave of the multi-env kernel: the weights as MFMA A operands.  v_mfma_f32_16x16x4_f32 accumulates its four
// products one after the other in k order with the rounding of fmaf (tools/ubench: a chain of them equals the chain
// acc = fmaf(w[k], x[k], acc), k ascending, bit for bit in 51 200 of 51 200 outputs), so Z^T[unit][env] = W . X^T evaluated as
// 16-unit x 16-env tiles gives every (unit, env) exactly the value of the per-environment fmaf chains of the other rollout kernels.
// Lane (r = lane % 16, q = lane / 16): A operand of tile t, k step ks = W[unit 16 t + r][k = 4 ks + q]; the tile's result registers
// i = 0..3 hold unit 16 t + 4 q + i of env r.
template <int OCT, int CIT>
struct TileRegs {
  static constexpr int K1 = 4 * (OCT > CIT ? OCT : CIT);
  float w1[4][K1];        // first layer (policy nets: 4 OCT k steps used | cost net: 4 CIT)
  float w2[4][16];        // second layer
  float wh[16];           // wave 0: Wa[action r][k = 4 ks + q] at [ks] (the other per-unit / per-action constants: MultiShared::cst)
  static constexpr int NSEL = (16 * (CIT > 0 ? CIT : 1) + WAVE - 1) / WAVE;
  int sel[NSEL];          // cost net: select_dim entries this lane prepares ...
  float plo[NSEL], phi[NSEL];        // ... their action bounds (ConstraintNet.clip_actions) ...
  double pmean[NSEL], pden[NSEL];    // ... and observation statistics (--cn_normalize: mean, sqrt(var + eps)): constants of the rollout
};

template <int OCT, int CIT>
__device__ __forceinline__ void load_pol_tiles(const PolLayout& L, const float* __restrict__ PT, TileRegs<OCT, CIT>& R, float* cst) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int net = w < 3 ? w : 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ju = 16 * t + r;
#pragma unroll
    for (int ks = 0; ks < 4 * OCT; ++ks) { const int k = 4 * ks + q; R.w1[t][ks] = (ju < L.H1 && k < L.O) ? PT[L.W1[net] + k * L.H1 + ju] : 0.f; }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { const int k = 4 * ks + q; R.w2[t][ks] = (ju < L.H2 && k < L.H1) ? PT[L.W2[net] + k * L.H2 + ju] : 0.f; }
  }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) { const int k = 4 * ks + q; R.wh[ks] = (w == 0 && r < L.A && k < L.H2) ? PT[L.Wa + k * L.A + r] : 0.f; }
  {
    const int j = lane;
    cst[CST_B1 + j] = j < L.H1 ? PT[L.b1[net] + j] : 0.f;
    cst[CST_B2 + j] = j < L.H2 ? PT[L.b2[net] + j] : 0.f;
    cst[CST_WO + j] = (w != 0 && j < L.H2) ? PT[(w == 1 ? L.Wv : L.Wc) + j] : 0.f;
    if (j < 16) {
      // constants of the whole rollout: evaluated once here instead of once per env step (same expressions as load_pol_regs)
      const int a = j < L.A ? j : 0;
      const float ls = (w == 0 && !L.discrete) ? PT[L.log_std + a] : 0.f;
      const float sd = expf(ls);
      cst[CST_BA + j] = w == 0 ? PT[L.ba + a] : 0.f;
      cst[CST_SD + j] = sd; cst[CST_LSD + j] = logf(sd); cst[CST_I2V + j] = 2.f * (sd * sd);
    }
    if (j == 0) cst[CST_BO] = w == 0 ? 0.f : PT[w == 1 ? L.bv : L.bc];
  }
}

template <int OCT, int CIT>
__device__ __forceinline__ void load_cn_tiles(const icrl_costnet_t& cn, const CnLayout& L, TileRegs<OCT, CIT>& R, float* cst) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* PT = cn.params_t;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ju = 16 * t + r;
#pragma unroll
    for (int ks = 0; ks < 4 * CIT; ++ks) { const int k = 4 * ks + q; R.w1[t][ks] = (ju < L.H1 && k < L.in) ? PT[L.W0 + k * L.H1 + ju] : 0.f; }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { const int k = 4 * ks + q; R.w2[t][ks] = (L.nh == 2 && ju < L.H2 && k < L.H1) ? PT[L.W1 + k * L.H2 + ju] : 0.f; }
  }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) R.wh[ks] = 0.f;
  {
    const int j = lane;
    cst[CST_B1 + j] = j < L.H1 ? PT[L.b0 + j] : 0.f;
    cst[CST_B2 + j] = (L.nh == 2 && j < L.H2) ? PT[L.b1 + j] : 0.f;
    cst[CST_WO + j] = j < L.H2 ? PT[L.Wo + j] : 0.f;
    if (j == 0) cst[CST_BO] = PT[L.bo];
  }
#pragma unroll
  for (int i = 0; i < (16 * CIT + WAVE - 1) / WAVE; ++i) {
    const int idx = lane + i * WAVE;
    const int sel = idx < L.in ? cn.select_dim[idx] : -1;
    R.sel[i] = sel;
    R.plo[i] = 0.f; R.phi[i] = 0.f; R.pmean[i] = 0.0; R.pden[i] = 1.0;
    if (sel >= 0 && sel < cn.obs_dim) {
      if (cn.obs_mean != nullptr && cn.obs_var != nullptr) { R.pmean[i] = cn.obs_mean[sel]; R.pden[i] = sqrt(cn.obs_var[sel] + cn.eps); }
    } else if (sel >= 0) {
      const int ai = sel - cn.obs_dim;
      if (cn.action_low != nullptr && cn.action_high != nullptr) { R.plo[i] = cn.action_low[ai]; R.phi[i] = cn.action_high[ai]; }
    }
  }
}

// wave_sum_fast's association for a value that lives as [tile t][i] = element 16 t + 4 q + i of env r: element c + 16 m sits in
// lane c + 16 m there, here c = 4 q + i and m = t.  wave_sum_fast: u[c] = (v[c] + v[c + 32]) + (v[c + 16] + v[c + 48]), then the
// 16 u's as ((u0 + u1) + (u2 + u3)) quads, quads pairwise, halves.  Returns the sum of env r in all four q lanes.
__device__ __forceinline__ float tile_sum64(const f32x4 (&v)[4]) {
  float u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (v[0][i] + v[2][i]) + (v[1][i] + v[3][i]);
  const float quad = (u[0] + u[1]) + (u[2] + u[3]);
  return xor32_sum(xor16_sum(quad));
}

static inline size_t multi_dyn_lds(int N, int O, int A, int owners) {
  // dynamics matrix | per owner wave: column buffer (N + 64) | ret / cost owners: done flags [2][N] bytes.  (The discounted returns of
  // the previous step stay in nm.ret / nm.cost_ret: only their owner wave touches them during the launch.)
  return ((size_t)O * A + (size_t)owners * (size_t)(N + 64)) * sizeof(double) + 2 * (size_t)((N + 15) / 16 * 16);
}

// synthetic env step of up to THREE environments at once by one wave (obs_dim <= 21): lane 21 s + i handles component i of the env in
// slot s.  em / n / mine are PER-LANE: the env's slot in the workgroup's LDS state, its index in the run, and whether the lane has an
// env at all.  The per-component arithmetic, the reward and the auto-reset draw are env_step_wave's, operation for operation; only
// the lanes they run on differ.  Leaves s_new, ctr, tep, rew, done of every handled env in `sh`.
template <int E, int CIT>
__device__ __noinline__ void env_step_wave3(int O, int A, int flags, int max_steps, const double __attribute__((address_space(3)))* B,
                                            double* s_all, int32_t* t_ep_all, uint32_t* step_count_all,
                                            MultiShared<E, CIT> __attribute__((address_space(3)))* shp,
                                            int em, int n, bool mine, int ci, int slot) {
  // (not inlined — inlined, the kernel spills; so everything comes in by value: the env's few scalars (flags = broken | reward_form << 1
  // | wall_terminate << 8), the workgroup's LDS state and the LDS copy of the dynamics matrix as LDS-address-space pointers, the env
  // arrays as global pointers — through a struct reference every access would be a flat_* instruction)
  MultiShared<E, CIT> __attribute__((address_space(3)))& sh = *shp;
  const bool broken = flags & 1, wall_terminate = (flags >> 8) & 1;
  const int reward_form = (flags >> 1) & 127;
  const bool live = mine && ci < O;
  double a[MAX_ACT];
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < MAX_ACT; ++j) {
    a[j] = 0.0;
    if (j < A) {
      a[j] = (double)sh.act_clip[em][j];
      if (broken && j >= 4) a[j] = 0.0;
      sq = sq + a[j] * a[j];
    }
  }
  const uint32_t ky = sh.key[em], ct = sh.ctr[em];
  const int i = ci < O ? ci : 0;
  double ns = 0.0;
  {
    double acc = 0.99 * sh.s_old[em][i];
    const double __attribute__((address_space(3)))* Bi = B + (size_t)i * A;
#pragma unroll
    for (int j = 0; j < MAX_ACT; ++j)
      if (j < A) acc = acc + Bi[j] * a[j];
    const double eps = (unit_uniform(ky, ct, (uint32_t)i) - 0.5) * 3.4641016151377544;
    ns = acc + 0.01 * eps;
  }
  const int base = 21 * (slot < 3 ? slot : 0);
  const double n0 = __shfl(ns, base, 64);
  const double n1 = __shfl(ns, base + 1, 64);
  double rw;
  if (reward_form == 0) rw = fabs(n0 - sh.s_old[em][0]) / 0.05 - 0.1 * sq;
  else rw = (sqrt(n0 * n0 + n1 * n1) + 1.0) - 0.5 * sq;
  int d = 0;
  if (wall_terminate && n0 <= -3.0) { rw = 0.0; d = 1; }
  const int tep = sh.tep[em] + 1;
  if (tep >= max_steps) d = 1;
  __builtin_amdgcn_wave_barrier();           // every lane has read ctr / tep / s_old before lane 0 of a slot replaces them
  if (live) {
    double v = ns;
    if (d) v = reward_form >= 2 ? -1.0 : (unit_uniform(ky, ct + 1u, (uint32_t)(O + i)) - 0.5) * 0.2;   // auto-reset draw (env_reset_value)
    as_global(s_all)[(size_t)n * O + i] = v;
    sh.s_new[em][i] = v;
  }
  if (mine && ci == 0) {
    const int tnew = d ? 0 : tep;
    sh.tep[em] = tnew; sh.ctr[em] = ct + 1u; sh.rew[em] = rw; sh.done[em] = d;
    as_global(t_ep_all)[n] = tnew; as_global(step_count_all)[n] = ct + 1u;
  }
}

template <int OCT, int CIT, int E>
__device__ __forceinline__ void rollout_multi_body(const WideArgs& p, const int g_arg) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  __shared__ MultiShared<E, CIT> sh;
  const ActStepArgs& a = p.act;
  // private copies of the structs whose pointers the step loop goes through, every pointer marked as a global-memory pointer
  // (common.h: as_global — in a batched launch the block comes from memory and the accesses would be flat_* otherwise)
  icrl_norm_t nm = p.nm; globalize(nm);
  icrl_buffer_t buf = a.buf; globalize(buf);
  icrl_agent_t ag = a.ag; globalize(ag);
  icrl_costnet_t cnet = a.cn; globalize(cnet);
  unsigned long long* const xg_all = as_global(p.xg);      // granules of every env, both parities
  unsigned long long* const sg_all = as_global(p.sg);      // statistics granules
  const float* const noise_g = as_global(a.noise);
  TileRegs<OCT, CIT> R;                // waves 0..2: policy / value / cost-value net, wave 3: cost net — as MFMA A operands
  if (threadIdx.x >= 192 && a.has_cn) load_cn_tiles<OCT, CIT>(a.cn, a.cl, R, sh.cst[3]);
  else load_pol_tiles<OCT, CIT>(a.pl, a.PT, R, sh.cst[threadIdx.x >> 6]);
  const int g = g_arg, G = p.G;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = a.pl.O, A = a.pl.A, N = a.env.n_envs, T = p.T;
  const int AS = buf.act_store;
  const int GX = 2 * O + 4, GS = 4 * O + 4;
  // Exchange records: 16 bytes {tag, lo, hi, tag} instead of two 8-byte {tag, word} granules per float64 (same bytes, half the memory
  // instructions; the done flag rides in the top bit of the reward record's second tag).  Per env R16 = obs + 2 records (obs columns,
  // reward, cost); statistics: (mean, var) per column, then the ret / cost denominators.
  const int R16 = O + 2;
  typedef unsigned int rec_u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xg_all, 0, 2 * N * GX * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(sg_all, 0, 2 * GS * 8, 0x00020000);
  // record stores: `sc1` (agent scope), or `sc0` when every workgroup of this run sits on this XCD (xcd_local, set below): the line then
  // stays in the XCD's L2 for the other workgroups' `sc1` loads instead of being dropped (ppo_common.h, "XCD placement"; measured at
  // HC x 64, 8 envs per workgroup: 12.2 -> 10.7 us per step).  Records carry their step tag at both ends: a stale read retries.
  bool xcd_local = false;
  auto rstore = [&](const __amdgpu_buffer_rsrc_t& rs, int byte_off, rec_u4 v) {
    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 1);      // sc0
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16);               // sc1
  };
  auto rload = [](const __amdgpu_buffer_rsrc_t& rs, int byte_off) -> rec_u4 { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); };
  const int Eg = (N - g + G - 1) / G;                // envs of this workgroup (1 .. E)
  const bool has_cost = a.has_cn != 0;
  const bool has_box = a.alow != nullptr && a.ahigh != nullptr;
  const int n_stats = O + (has_cost ? 2 : 1);
  const int owners = (n_stats + G - 1) / G;          // owner waves per workgroup (<= 4: checked by the launcher)
  double* const Bl = dyn_lds;
  double* const colbuf = Bl + O * a.env.act_dim + (size_t)w * (N + 64);      // this wave's column buffer (owner waves only)
  unsigned char* const done_bufs = reinterpret_cast<unsigned char*>(Bl + O * a.env.act_dim + (size_t)owners * (N + 64));
  icrl_env_t env = a.env; globalize(env);
  for (int i = tid; i < O * a.env.act_dim; i += 256) Bl[i] = a.env.B[i];
  env.B = Bl;
  const int env_flags = (env.broken ? 1 : 0) | (env.reward_form << 1) | (env.wall_terminate ? 256 : 0);
  if (tid < MAX_ACT) { sh.alow[tid] = (has_box && tid < A) ? a.alow[tid] : 0.f; sh.ahigh[tid] = (has_box && tid < A) ? a.ahigh[tid] : 0.f; }
  for (int idx = tid; idx < E * MULTI_OP; idx += 256) {
    const int e = idx / MULTI_OP, i = idx % MULTI_OP;
    const bool live = e < Eg && i < O;
    const int n = g + e * G;
    const double lo = live ? ag.last_obs[(size_t)n * O + i] : 0.0;
    sh.x[e][i] = (float)lo;
    sh.olast[e][i] = lo;
    sh.s_old[e][i] = live ? a.env.s[(size_t)n * O + i] : 0.0;
    sh.s_new[e][i] = 0.0;
  }
  if (tid < E) {
    const int e = tid, n = g + (e < Eg ? e : 0) * G;
    sh.ctr[e] = a.env.step_count[n]; sh.tep[e] = a.env.t_ep[n]; sh.last_done[e] = ag.last_dones[n]; sh.key[e] = a.env.key[n];
    sh.rew[e] = 0.0; sh.cost[e] = 0.f; sh.done[e] = 0;
  }
  // the statistic this WAVE owns (if any): sid = g + w * G
  const int sid = g + w * G;
  const bool owner = w < owners && sid < n_stats;
  const int own_col = (owner && sid < O) ? sid : -1;
  const bool own_ret = owner && sid == O, own_cost = owner && has_cost && sid == O + 1;
  double* const retbuf = own_cost ? nm.cost_ret : nm.ret;           // [N] in global memory: this wave is its only user during the launch
  unsigned char* const done_all = done_bufs + (own_cost ? (size_t)((N + 15) / 16 * 16) : 0);
  double o_mean = 0.0, o_var = 1.0, o_cnt = 0.0;
  if (own_col >= 0) { o_mean = nm.obs_mean[own_col]; o_var = nm.obs_var[own_col]; o_cnt = nm.obs_count[0]; }
  if (own_ret) { o_mean = nm.ret_stats[0]; o_var = nm.ret_stats[1]; o_cnt = nm.ret_stats[2]; }
  if (own_cost) { o_mean = nm.cost_stats[0]; o_var = nm.cost_stats[1]; o_cnt = nm.cost_stats[2]; }
  const int NA = A;                                  // (continuous actions only: the launcher keeps discrete policies elsewhere)
  if (tid < E * MAX_ACT) {
    const int e = tid / MAX_ACT, k = tid % MAX_ACT;
    sh.noise[0][e][k] = (e < Eg && k < NA) ? noise_g[((size_t)(g + e * G)) * NA + k] : 0.f;
  }
  const int r16 = lane & 15, q4 = lane >> 4;          // MFMA lane coordinates: env row r16, k / unit quarter q4
  const int er = r16 < E ? r16 : E - 1;               // rows beyond E replicate the last env's row (their results are not used)
  const int nk1 = (O + 3) / 4, nkc = a.has_cn ? (a.cl.in + 3) / 4 : 0;
  const int pH1 = a.pl.H1, pH2 = a.pl.H2, cH1 = a.cl.H1, cH2 = a.cl.H2, cnh = a.cl.nh;      // (read once: the block may live in LDS)
  const bool cn_norm = a.has_cn && cnet.obs_mean != nullptr && cnet.obs_var != nullptr;
  const bool cn_box = a.has_cn && cnet.action_low != nullptr && cnet.action_high != nullptr;
  const double cn_clip = cnet.clip_obs;
  int spin_limit = 1 << 22;
  if (p.xcc != nullptr && tid == 0) sh.xcd_local = all_on_one_xcd(as_global(p.xcc), 1, g, G) ? 1 : 0;
  __syncthreads();
  if (p.xcc != nullptr) xcd_local = __builtin_amdgcn_readfirstlane(sh.xcd_local) != 0;
  const bool prof = p.prof != 0 && g == p.prof - 1 && blockIdx.y == 0;
  unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, pcA = 0, pcB = 0, tl = prof ? prof_now() : 0ull;
  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const unsigned gtag = (unsigned)(t + 1);
    float noise_next = 0.f;
    const int pe = tid / MAX_ACT, pk = tid % MAX_ACT;
    const bool pf = tid < E * MAX_ACT && pe < Eg && pk < NA && t + 1 < T;
    if (pf) noise_next = noise_g[((size_t)(t + 1) * N + g + pe * G) * NA + pk];
    // ---------------- phase A: the three MLPs for all E envs at once, as MFMA tiles (TileRegs) ----------------
    // A wave consumes only its own network's activations: the hand-over between layers (result layout [unit 16 t + 4 q + i][env r]
    // -> B operand layout [k = 4 ks + q][env r]) goes through the wave's own LDS rows, no workgroup barrier until the heads are done.
    if (w < 3) {
      f32x4 z[4], cb[4];
      float bop[16];
      const float* const cw = &sh.cst[w][4 * q4];       // this lane's slice of the wave's constants: + CST_x + 16 t
#pragma unroll
      for (int t = 0; t < 4; ++t) { z[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[t] = *reinterpret_cast<const f32x4*>(cw + CST_B1 + 16 * t); }
      {
        const float* xb = &sh.x[er][q4];
        float bx[4 * OCT];              // every B operand of the layer in flight before the first MFMA (pad columns are zero)
#pragma unroll
        for (int ks = 0; ks < 4 * OCT; ++ks) bx[ks] = xb[4 * ks];
#pragma unroll
        for (int ks = 0; ks < 4 * OCT; ++ks) {
          if (ks < nk1) {               // (k steps that only meet padding — zero weights — are not issued)
#pragma unroll
            for (int t = 0; t < 4; ++t) z[t] = MFMA16(R.w1[t][ks], bx[ks], z[t]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < pH1 ? fast_tanh(z[t][i] + cb[t][i]) : 0.f;
      if (r16 < E) {
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&sh.h[w][r16][16 * t + 4 * q4]) = z[t];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      {
        const float* hb = &sh.h[w][er][q4];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bop[ks] = hb[4 * ks];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { z[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[t] = *reinterpret_cast<const f32x4*>(cw + CST_B2 + 16 * t); }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) z[t] = MFMA16(R.w2[t][ks], bop[ks], z[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < pH2 ? fast_tanh(z[t][i] + cb[t][i]) : 0.f;
      if (w == 0) {                     // action head + Gaussian sample / log-prob of every env: lane (r, q) holds actions 4 q + i of env r
        __builtin_amdgcn_wave_barrier();        // (every lane's reads of h are consumed: the MFMAs above used them)
        if (r16 < E) {
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&sh.h[0][r16][16 * t + 4 * q4]) = z[t];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        {
          const float* hb = &sh.h[0][er][q4];
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) bop[ks] = hb[4 * ks];
        }
        f32x4 m = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) m = MFMA16(R.wh[ks], bop[ks], m);
        const f32x4 nz = *reinterpret_cast<const f32x4*>(&sh.noise[par][er][4 * q4]);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&sh.alow[4 * q4]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(&sh.ahigh[4 * q4]);
        const f32x4 hb_ = *reinterpret_cast<const f32x4*>(cw + CST_BA), sd_ = *reinterpret_cast<const f32x4*>(cw + CST_SD);
        const f32x4 lsd_ = *reinterpret_cast<const f32x4*>(cw + CST_LSD), i2v_ = *reinterpret_cast<const f32x4*>(cw + CST_I2V);
        f32x4 araw, aclip;
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mean = m[i] + hb_[i];
          float lp = 0.f, act = 0.f, c = 0.f;
          if (4 * q4 + i < A) {
            act = mean + nz[i] * sd_[i];                  // Normal.rsample: loc + eps * scale
            const float diff = act - mean;
            lp = -(diff * diff) / i2v_[i] - lsd_[i] - LOG_SQRT_2PI_F;
            c = act;
            if (has_box) c = fminf(fmaxf(act, lo[i]), hi[i]);
          }
          araw[i] = act; aclip[i] = c;
          u[i] = (lp + 0.f) + (0.f + 0.f);                // wave_sum_fast over lane = action: lanes 16.. hold 0
        }
        const float lp_env = xor32_sum(xor16_sum((u[0] + u[1]) + (u[2] + u[3])));
        if (r16 < E) {
          *reinterpret_cast<f32x4*>(&sh.act_raw[r16][4 * q4]) = araw;
          *reinterpret_cast<f32x4*>(&sh.act_clip[r16][4 * q4]) = aclip;
          if (q4 == 0) sh.scal[r16][2] = lp_env;
        }
      } else {                          // value heads: products with the output weights, summed in wave_sum_fast's order
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 wo = *reinterpret_cast<const f32x4*>(cw + CST_WO + 16 * t);
#pragma unroll
          for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < pH2 ? wo[i] * z[t][i] : 0.f;
        }
        const float v = tile_sum64(z) + sh.cst[w][CST_BO];
        if (r16 < E && q4 == 0) sh.scal[r16][w - 1] = v;
      }
    }
    __syncthreads();
    if (prof) { const unsigned long long tn_ = prof_now(); pc0 += tn_ - tl; tl = tn_; }     // the three MLPs + heads
    // ---------------- env steps (waves 0..2), cost net (wave 3), buffer rows ----------------
    if (OCT <= 2 && w < 3 && O <= 21 && env.reward_form < 2) {
      // narrow observations: THREE envs per pass, lane 21 s + i = component i of slot s (env 3 q + s; triple q on wave q % 3)
      const int slot = lane / 21, ci = lane - 21 * slot;
      for (int q = w; 3 * q < Eg; q += 3) {
        const int e = 3 * q + slot;
        const bool mine = slot < 3 && e < Eg;
        const int em = mine ? e : 3 * q;
        const int n = g + em * G;
        const size_t tn = (size_t)t * N + n;
        const int xrec = (par * N + n) * R16 * 16;      // byte offset of the env's records of this parity
        if (mine) {       // rows that depend on the pre-step state only
          if (ci < O) { buf.observations[tn * O + ci] = sh.x[em][ci]; buf.orig_observations[tn * O + ci] = (float)sh.s_old[em][ci]; }
          if (ci < AS) buf.actions[tn * AS + ci] = sh.act_raw[em][ci];
          if (ci < A) ag.act_clipped[(size_t)n * A + ci] = sh.act_clip[em][ci];
          if (ci == 0) {
            buf.dones[tn] = (float)sh.last_done[em];
            buf.reward_values[tn] = sh.scal[em][0];
            buf.cost_values[tn] = sh.scal[em][1];
            buf.log_probs[tn] = sh.scal[em][2];
            ag.last_v_r[n] = sh.scal[em][0];
            ag.last_v_c[n] = sh.scal[em][1];
          }
        }
        env_step_wave3<E, CIT>(O, env.act_dim, env_flags, env.max_steps, (const double __attribute__((address_space(3)))*)Bl, env.s, env.t_ep, env.step_count,
                               (MultiShared<E, CIT> __attribute__((address_space(3)))*)&sh, em, n, mine, ci, slot);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // the wave's LDS writes (s_new, rew, done) are visible to its own reads
        if (mine) {
          if (ci < O) {
            const double v = sh.s_new[em][ci];
            buf.new_orig_observations[tn * O + ci] = (float)v;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            rstore(xrs, xrec + 16 * ci, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag});
          }
          if (ci == 0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sh.rew[em]);
            rstore(xrs, xrec + 16 * O, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag | (sh.done[em] ? 0x80000000u : 0u)});
          }
        }
      }
    } else if (w < 3) {
      for (int e = w; e < Eg; e += 3) {
        const int n = g + e * G;
        const size_t tn = (size_t)t * N + n;
        const int xrec = (par * N + n) * R16 * 16;      // byte offset of the env's records of this parity
        // rows that depend on the pre-step state only
        float* ob = buf.observations + tn * O;
        float* oob = buf.orig_observations + tn * O;
        for (int i = lane; i < O; i += WAVE) { ob[i] = sh.x[e][i]; oob[i] = (float)sh.s_old[e][i]; }
        if (lane < AS) buf.actions[tn * AS + lane] = sh.act_raw[e][lane];
        if (lane < A) ag.act_clipped[(size_t)n * A + lane] = sh.act_clip[e][lane];
        if (lane == 0) {
          buf.dones[tn] = (float)sh.last_done[e];
          buf.reward_values[tn] = sh.scal[e][0];
          buf.cost_values[tn] = sh.scal[e][1];
          buf.log_probs[tn] = sh.scal[e][2];
          ag.last_v_r[n] = sh.scal[e][0];
          ag.last_v_c[n] = sh.scal[e][1];
        }
        double rew; int done;
        uint32_t e_ctr = sh.ctr[e];
        int e_tep = sh.tep[e];
        env_step_wave(env, n, sh.s_old[e], sh.act_clip[e], sh.key[e], e_ctr, e_tep, sh.s_new[e], rew, done);
        float* nob = buf.new_orig_observations + tn * O;
        for (int i = lane; i < O; i += WAVE) {
          const double v = sh.s_new[e][i];
          nob[i] = (float)v;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
          rstore(xrs, xrec + 16 * i, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag});
        }
        if (lane == 0) {
          const unsigned long long bits = (unsigned long long)__double_as_longlong(rew);
          rstore(xrs, xrec + 16 * O, rec_u4{gtag, (unsigned)bits, (unsigned)(bits >> 32), gtag | (done ? 0x80000000u : 0u)});
          sh.ctr[e] = e_ctr; sh.tep[e] = e_tep; sh.rew[e] = rew; sh.done[e] = done;
        }
      }
    } else if (has_cost) {
      // cost net of all E envs at once (wave 3): prepare() per env into cx, then the ReLU layers as MFMA tiles
#pragma unroll
      for (int i = 0; i < (16 * CIT + WAVE - 1) / WAVE; ++i) {
        // this lane's input component idx = lane + 64 i of EVERY env: the loads of all envs first (no branch between them), then the
        // arithmetic of prepare() — operation for operation cost_forward_wave's
        const int idx = lane + i * WAVE;
        const int sel = R.sel[i];
        const bool is_obs = sel >= 0 && sel < cnet.obs_dim, is_act = sel >= cnet.obs_dim;
        const int so = is_obs ? sel : 0, sa = is_act ? sel - cnet.obs_dim : 0;
        double ov[E];
        float av[E], a0[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { ov[e] = sh.s_old[e][so]; av[e] = sh.act_clip[e][sa]; a0[e] = sh.act_clip[e][0]; }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          float v = 0.f;
          if (is_obs) {
            double o = ov[e];
            if (cn_norm) o = (o - R.pmean[i]) / R.pden[i];
            if (cn_clip >= 0.0) o = fmin(fmax(o, -cn_clip), cn_clip);
            v = (float)o;
          } else if (is_act) {
            float x;
            if (cnet.is_discrete) x = ((int)a0[e] == sa) ? 1.f : 0.f;
            else x = av[e];
            if (cn_box) x = fminf(fmaxf(x, R.plo[i]), R.phi[i]);
            v = x;
          }
          if (idx < 16 * CIT) sh.cx[e][idx] = v;       // pad entries are written as 0
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if (prof) pcA += prof_now() - tl;                 // (cost wave: inputs prepared)
      const int erc = r16 < Eg ? r16 : Eg - 1;          // (rows beyond the workgroup's envs replicate the last one)
      f32x4 z[4], cb[4];
      const float* const cw = &sh.cst[3][4 * q4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { z[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[t] = *reinterpret_cast<const f32x4*>(cw + CST_B1 + 16 * t); }
      {
        const float* xb = &sh.cx[erc][q4];
        float bx[4 * CIT];
#pragma unroll
        for (int ks = 0; ks < 4 * CIT; ++ks) bx[ks] = xb[4 * ks];
#pragma unroll
        for (int ks = 0; ks < 4 * CIT; ++ks) {
          if (ks < nkc) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (16 * t < cH1) z[t] = MFMA16(R.w1[t][ks], bx[ks], z[t]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < cH1 ? fmaxf(z[t][i] + cb[t][i], 0.f) : 0.f;
      if (cnh == 2) {
        if (r16 < E) {
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&sh.h[3][r16][16 * t + 4 * q4]) = z[t];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        float bop[16];
        {
          const float* hb = &sh.h[3][er][q4];
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) bop[ks] = hb[4 * ks];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { z[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[t] = *reinterpret_cast<const f32x4*>(cw + CST_B2 + 16 * t); }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (4 * ks < cH1) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (16 * t < cH2) z[t] = MFMA16(R.w2[t][ks], bop[ks], z[t]);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < cH2 ? fmaxf(z[t][i] + cb[t][i], 0.f) : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 wo = *reinterpret_cast<const f32x4*>(cw + CST_WO + 16 * t);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < cH2 ? wo[i] * z[t][i] : 0.f;
      }
      if (prof) pcB += prof_now() - tl;                 // (cost wave: hidden layers done)
      const float zz = tile_sum64(z) + sh.cst[3][CST_BO];
      const float zeta = 1.f / (1.f + expf(-zz));
      const float cost = 1.f - zeta;
      if (q4 == 0 && r16 < Eg) {
        const int n = g + r16 * G;
        rstore(xrs, ((par * N + n) * R16 + O + 1) * 16, rec_u4{gtag, __float_as_uint(cost), 0u, gtag});
        buf.orig_costs[(size_t)t * N + n] = cost;
        sh.cost[r16] = cost;
      }
    } else {
      if (lane < Eg) {
        const int n = g + lane * G;
        rstore(xrs, ((par * N + n) * R16 + O + 1) * 16, rec_u4{gtag, __float_as_uint(0.f), 0u, gtag});
        buf.orig_costs[(size_t)t * N + n] = 0.f;
        sh.cost[lane] = 0.f;
      }
    }
    if (prof) { const unsigned long long tn_ = prof_now(); pc1 += tn_ - tl; tl = tn_; }     // env steps / cost net + rows (this wave's part)
    // ---------------- phase B1: owner waves gather their statistic from all envs and publish it ----------------
    if (own_col >= 0 || own_ret || own_cost) {
      const int slot = own_col >= 0 ? own_col : (own_ret ? O : O + 1);      // the record of every env this owner gathers
      const bool wide = !own_cost;                                   // a float64 in the record, or the float32 cost
      const int gbase = par * N * R16 * 16;
      for (int i0 = 0; i0 < N; i0 += 4 * WAVE) {
        rec_u4 g0[4], gd[4];          // gd: the reward record of the env — the cost owner takes the done flag from its tag
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { g0[k] = gd[k] = rec_u4{0u, 0u, 0u, 0u}; ok[k] = i0 + k * WAVE + lane >= N; }
        double rprev[4] = {0.0, 0.0, 0.0, 0.0};      // previous discounted returns (return owners): in flight while the granules are polled
        if (own_col < 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { const int i = i0 + k * WAVE + lane; if (i < N) rprev[k] = retbuf[i]; }
        }
        for (int spins = 0; spins < spin_limit; ++spins) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * WAVE + lane;
            if (!ok[k]) { g0[k] = rload(xrs, gbase + (i * R16 + slot) * 16); gd[k] = own_cost ? rload(xrs, gbase + (i * R16 + O) * 16) : g0[k]; }
          }
          bool all = true;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (!ok[k]) ok[k] = g0[k][0] == gtag && (g0[k][3] & 0x7fffffffu) == gtag && gd[k][0] == gtag && (gd[k][3] & 0x7fffffffu) == gtag;
            all = all && ok[k];
          }
          if (__all(all)) break;
          if (spins + 1 == spin_limit) spin_limit = 1;
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * WAVE + lane;
          if (i < N) {
            double v;
            if (wide) v = __longlong_as_double((long long)(((unsigned long long)g0[k][2] << 32) | (unsigned long long)g0[k][1]));
            else v = (double)__uint_as_float(g0[k][1]);
            if (own_col >= 0) colbuf[i] = v;
            else {
              colbuf[i] = rprev[k] * (own_ret ? nm.reward_gamma : nm.cost_gamma) + v;      // vec_normalize.py:102, 245
              done_all[i] = (unsigned char)(gd[k][3] >> 31);
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      double bm, bv;
      if (own_col >= 0) {
        column_moments_contig(colbuf, N, bm, bv);                   // every lane computes the same chain (numpy's axis-0 order)
      } else {
        bm = np_pairwise_sum_wave(colbuf, N) / (double)N;           // numpy's 1-D pairwise order
        // returns of finished episodes restart at 0 (vec_normalize.py:99, 241); then the squared deviations in place
        for (int i = lane; i < N; i += WAVE) { const double c = colbuf[i]; retbuf[i] = done_all[i] ? 0.0 : c; const double d = c - bm; colbuf[i] = d * d; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        bv = np_pairwise_sum_wave(colbuf, N) / (double)N;
      }
      chan_merge(o_mean, o_var, o_cnt, bm, bv, (double)N);
      o_cnt = (double)N + o_cnt;
      const int sbase = par * (GS / 2) * 16;          // statistics records of this parity: (mean, var) per column, then the two denominators
      if (lane == 0) {
        if (own_col >= 0) {
          const unsigned long long mb = (unsigned long long)__double_as_longlong(o_mean), vb = (unsigned long long)__double_as_longlong(o_var);
          rstore(srs, sbase + 32 * own_col, rec_u4{gtag, (unsigned)mb, (unsigned)(mb >> 32), gtag});
          rstore(srs, sbase + 32 * own_col + 16, rec_u4{gtag, (unsigned)vb, (unsigned)(vb >> 32), gtag});
        } else {
          const unsigned long long db = (unsigned long long)__double_as_longlong(sqrt(o_var + nm.epsilon));
          rstore(srs, sbase + (2 * O + (own_ret ? 0 : 1)) * 16, rec_u4{gtag, (unsigned)db, (unsigned)(db >> 32), gtag});
        }
      }
    }
    if (prof) { const unsigned long long tn_ = prof_now(); pc2 += tn_ - tl; tl = tn_; }     // owner: gather + moments + publish
    // ---------------- phase B2: everybody reads the statistics granules, then normalises its envs ----------------
    {
      const int sbase = par * (GS / 2) * 16;
      const int total = has_cost ? GS / 2 : GS / 2 - 1;              // records: (mean, var) per column, ret denominator, cost denominator
      for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + tid;
        rec_u4 gv = rec_u4{gtag, 0u, 0u, gtag};
        bool ok = i >= total;
        for (int spins = 0; spins < spin_limit && !ok; ++spins) {
          gv = rload(srs, sbase + i * 16);
          ok = gv[0] == gtag && gv[3] == gtag;
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) spin_limit = 1;
        if (i < total) {
          const double v = __longlong_as_double((long long)(((unsigned long long)gv[2] << 32) | (unsigned long long)gv[1]));
          if (i < 2 * O) { if (i & 1) sh.var[i >> 1] = v; else sh.mean[i >> 1] = v; }
          else sh.dens[i - 2 * O] = v;
        }
      }
    }
    __syncthreads();
    if (prof) { const unsigned long long tn_ = prof_now(); pc3 += tn_ - tl; tl = tn_; }     // statistics granules arrived (+ barrier)
    for (int idx = tid; idx < Eg * O; idx += 256) {      // (env, component) pairs spread over all threads: one float64 sqrt + division each
      const int e = idx / O, i = idx - e * O;
      {
        const size_t tn = (size_t)t * N + g + e * G;
        const double raw = sh.s_new[e][i];
        double o = raw;
        if (nm.norm_obs) o = fmin(fmax((o - sh.mean[i]) / sqrt(sh.var[i] + nm.epsilon), -nm.clip_obs), nm.clip_obs);
        sh.olast[e][i] = o;
        sh.x[e][i] = (float)o;
        buf.new_observations[tn * O + i] = (float)o;
        sh.s_old[e][i] = raw;
      }
    }
    if (tid < Eg) {
      const int e = tid;
      const size_t tn = (size_t)t * N + g + e * G;
      double r = sh.rew[e];
      if (nm.norm_reward) r = fmin(fmax(r / sh.dens[0], -nm.clip_reward), nm.clip_reward);
      buf.rewards[tn] = (float)r;
      if (has_cost) {
        double c = (double)sh.cost[e];
        if (nm.norm_cost) c = fmin(fmax(c / sh.dens[1], -nm.clip_cost), nm.clip_cost);
        buf.costs[tn] = (float)c;
      }
      sh.last_done[e] = sh.done[e];
    }
    if (pf) sh.noise[par ^ 1][pe][pk] = noise_next;
    __syncthreads();
    if (prof) { const unsigned long long tn_ = prof_now(); pc4 += tn_ - tl; tl = tn_; }     // normalise + rows
  }
  if (prof && (tid == 0 || tid == 192)) {      // wave 0: a policy / env wave's view; wave 3: the cost wave's
    unsigned long long* o = g_rollout_prof_wide + (tid == 192 ? 8 : 0);
    o[0] = pc0; o[1] = pc1; o[2] = pc2; o[3] = pc3; o[4] = pc4; o[5] = (unsigned long long)T; o[6] = pcA; o[7] = pcB;
  }
  if (spin_limit == 1 && ag.status != nullptr && (tid & 63) == 0) atomicOr(ag.status, 1);
  // ---- leave the agent / wrapper / normaliser state exactly where the per-step path leaves it
  for (int idx = tid; idx < Eg * MULTI_OP; idx += 256) {
    const int e = idx / MULTI_OP, i = idx % MULTI_OP;
    if (i < O) ag.last_obs[(size_t)(g + e * G) * O + i] = sh.olast[e][i];
  }
  if (tid < Eg) {
    const int e = tid, n = g + e * G;
    ag.last_dones[n] = (uint8_t)sh.last_done[e];
    ag.raw_rew[n] = sh.rew[e]; ag.dones[n] = (uint8_t)sh.done[e];
    if (has_cost) ag.raw_cost[n] = sh.cost[e];
  }
  if (lane == 0) {
    if (own_col >= 0) { nm.obs_mean[own_col] = o_mean; nm.obs_var[own_col] = o_var; if (own_col == 0) nm.obs_count[0] = o_cnt; }
    if (own_ret) { nm.ret_stats[0] = o_mean; nm.ret_stats[1] = o_var; nm.ret_stats[2] = o_cnt; }
    if (own_cost) { nm.cost_stats[0] = o_mean; nm.cost_stats[1] = o_var; nm.cost_stats[2] = o_cnt; }
  }
}

template <int OCT, int CIT, int E>
__global__ void __launch_bounds__(256) rollout_multi_kernel(WideArgs p, int packed) {
  if (packed && (blockIdx.x & 7)) return;      // the G <= 32 workgroups on one XCD: workgroups 0, 8, 16, ... (ppo_common.h, "XCD placement")
  rollout_multi_body<OCT, CIT, E>(p, packed ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
}

// several independent runs in ONE launch: grid (G, n_runs), run = blockIdx.y
template <int OCT, int CIT, int E>
__global__ void __launch_bounds__(256) rollout_multi_batch_kernel(const WideArgs* __restrict__ runs, int n_runs, int G, int packed) {
  __shared__ WideArgs p;
  int run = (int)blockIdx.y, g = (int)blockIdx.x;
  if (packed) {      // the G workgroups of a run on one XCD: workgroups b, b + 8, ... (ppo_common.h, "XCD placement")
    const int id = (int)blockIdx.x, per = 8 * G;
    run = (id / per) * 8 + (id & 7);
    g = (id % per) >> 3;
    if (run >= n_runs) return;
  }
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(runs + run);
    unsigned* dst = reinterpret_cast<unsigned*>(&p);
    for (unsigned i = threadIdx.x; i < sizeof(WideArgs) / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  rollout_multi_body<OCT, CIT, E>(p, g);
}

// VecNormalizeWithCost.reset (vec_normalize.py:148-157, 270-278)
__global__ void __launch_bounds__(1024) norm_reset_kernel(icrl_norm_t nm, const double* raw_obs, int N, int O,
                                                          double* obs_out) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int n = tid; n < N; n += nt) { nm.ret[n] = 0.0; nm.cost_ret[n] = 0.0; }
  __syncthreads();
  if (nm.training && tid == 0) {
    // ret = ret * gamma + ret = 0 -> update with a zero batch: mean 0, var 0, count N
    double m = nm.ret_stats[0], v = nm.ret_stats[1];
    chan_merge(m, v, nm.ret_stats[2], 0.0, 0.0, (double)N);
    nm.ret_stats[0] = m; nm.ret_stats[1] = v; nm.ret_stats[2] = (double)N + nm.ret_stats[2];
    m = nm.cost_stats[0]; v = nm.cost_stats[1];
    chan_merge(m, v, nm.cost_stats[2], 0.0, 0.0, (double)N);
    nm.cost_stats[0] = m; nm.cost_stats[1] = v; nm.cost_stats[2] = (double)N + nm.cost_stats[2];
  }
  for (int idx = tid; idx < N * O; idx += nt) {
    const int j = idx % O;
    double o = raw_obs[idx];
    if (nm.norm_obs) o = fmin(fmax((o - nm.obs_mean[j]) / sqrt(nm.obs_var[j] + nm.epsilon), -nm.clip_obs), nm.clip_obs);
    obs_out[idx] = o;
  }
}

// =================================================================================================================
// fine-grained kernels behind the per-call C ABI
// =================================================================================================================
template <int OCT>
__global__ void __launch_bounds__(192) policy_forward_kernel(PolLayout pl, const float* PT, const double* obs,
                                                             const float* noise, int deterministic, const float* alow,
                                                             const float* ahigh, float* actions, float* act_clipped,
                                                             float* v_r, float* v_c, float* log_prob,
                                                             const float* given, float* entropy) {
  __shared__ ActShared sh;
  PolRegs<OCT> R;
  load_pol_regs<OCT>(pl, PT, R);
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < MAX_OBS; i += 192) sh.x[i] = i < pl.O ? (float)obs[(size_t)n * pl.O + i] : 0.f;
  __syncthreads();
  const float* noise_row = noise ? noise + (size_t)n * (pl.discrete ? 1 : pl.A) : nullptr;
  const int AS = pl.discrete ? 1 : pl.A;
  policy_forward_block<OCT>(pl, R, sh, noise_row, deterministic || noise == nullptr, alow, ahigh,
                            given ? given + (size_t)n * AS : nullptr);
  __syncthreads();
  if (tid < AS) {
    if (actions) actions[(size_t)n * AS + tid] = sh.act_raw[tid];
    if (act_clipped) act_clipped[(size_t)n * AS + tid] = sh.act_clip[tid];
  }
  if (tid == 0) {
    if (v_r) v_r[n] = sh.scal[0];
    if (v_c) v_c[n] = sh.scal[1];
    if (log_prob) log_prob[n] = sh.scal[2];
    if (entropy) entropy[n] = sh.scal[3];
  }
}

// The same forward for MANY rows (policy.evaluate_actions over the nominal / expert samples of the KL metrics, forward over a batch):
// 16 rows per workgroup pass as fp32 MFMA tiles, exactly as rollout_multi_kernel evaluates 16 environments — every (unit, row) gets the
// value of policy_forward_block's fmaf chain (TileRegs), the heads and the log-probability / entropy sums are taken in
// wave_sum_fast's association.  Continuous actions only.  Waves: pi | vf | cvf; a workgroup walks tiles blockIdx.x, + gridDim.x, ...
template <int OCT>
__global__ void __launch_bounds__(192) policy_rows_kernel(PolLayout pl, const float* __restrict__ PT, const double* __restrict__ obs,
                                                          const float* __restrict__ noise, int deterministic, const float* __restrict__ alow,
                                                          const float* __restrict__ ahigh, float* actions, float* act_clipped, float* v_r,
                                                          float* v_c, float* log_prob, const float* __restrict__ given, float* entropy, int N) {
  constexpr int XS = 16 * OCT + 4;                       // row strides of 4 mod 32 floats: conflict-free B-operand reads
  __shared__ __attribute__((aligned(16))) float x[16][XS];
  __shared__ __attribute__((aligned(16))) float hbuf[3][16][MULTI_HS];
  __shared__ __attribute__((aligned(16))) float cst[3][MULTI_CST];
  TileRegs<OCT, 1> R;
  load_pol_tiles<OCT, 1>(pl, PT, R, cst[threadIdx.x >> 6]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int O = pl.O, A = pl.A, H1 = pl.H1, H2 = pl.H2;
  const int nk1 = (O + 3) / 4;
  const bool has_box = alow != nullptr && ahigh != nullptr;
  const bool sample = given == nullptr && !deterministic && noise != nullptr;
  for (int i = tid; i < 16 * XS; i += 192) (&x[0][0])[i] = 0.f;          // pad columns stay zero
  const int n_tiles = (N + 15) / 16;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int n0 = 16 * tile;
    const int rows = N - n0 < 16 ? N - n0 : 16;
    __syncthreads();                                      // the previous tile's x is consumed (first pass: the zero fill, cst)
    for (int i = tid; i < rows * O; i += 192) { const int rr = i / O, k = i - rr * O; x[rr][k] = (float)obs[(size_t)(n0 + rr) * O + k]; }
    __syncthreads();
    const int er = r16 < rows ? r16 : rows - 1;           // rows beyond the tile replicate its last row (their results are not stored)
    f32x4 z[4], cb[4];
    float bop[16];
    const float* const cw = &cst[w][4 * q4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { z[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[t] = *reinterpret_cast<const f32x4*>(cw + CST_B1 + 16 * t); }
    {
      const float* xb = &x[er][q4];
      float bx[4 * OCT];
#pragma unroll
      for (int ks = 0; ks < 4 * OCT; ++ks) bx[ks] = xb[4 * ks];
#pragma unroll
      for (int ks = 0; ks < 4 * OCT; ++ks) {
        if (ks < nk1) {
#pragma unroll
          for (int t = 0; t < 4; ++t) z[t] = MFMA16(R.w1[t][ks], bx[ks], z[t]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < H1 ? fast_tanh(z[t][i] + cb[t][i]) : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&hbuf[w][r16][16 * t + 4 * q4]) = z[t];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    {
      const float* hb = &hbuf[w][r16][q4];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bop[ks] = hb[4 * ks];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { z[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[t] = *reinterpret_cast<const f32x4*>(cw + CST_B2 + 16 * t); }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) z[t] = MFMA16(R.w2[t][ks], bop[ks], z[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < H2 ? fast_tanh(z[t][i] + cb[t][i]) : 0.f;
    const int n = n0 + r16;
    const bool live = r16 < rows;
    if (w == 0) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&hbuf[0][r16][16 * t + 4 * q4]) = z[t];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      {
        const float* hb = &hbuf[0][r16][q4];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bop[ks] = hb[4 * ks];
      }
      f32x4 m = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) m = MFMA16(R.wh[ks], bop[ks], m);
      const f32x4 hb_ = *reinterpret_cast<const f32x4*>(cw + CST_BA), sd_ = *reinterpret_cast<const f32x4*>(cw + CST_SD);
      const f32x4 lsd_ = *reinterpret_cast<const f32x4*>(cw + CST_LSD), i2v_ = *reinterpret_cast<const f32x4*>(cw + CST_I2V);
      float u[4], ue[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ai = 4 * q4 + i;
        const float mean = m[i] + hb_[i];
        float lp = 0.f, entl = 0.f;
        if (ai < A) {
          const size_t at = (size_t)(live ? n : n0) * A + ai;
          float act = mean;
          if (given != nullptr) act = given[at];
          else if (sample) act = mean + noise[at] * sd_[i];             // Normal.rsample: loc + eps * scale
          const float diff = act - mean;
          lp = -(diff * diff) / i2v_[i] - lsd_[i] - LOG_SQRT_2PI_F;
          entl = HALF_LOG_2PI_PLUS_HALF_F + lsd_[i];
          if (live) {
            if (actions) actions[at] = act;
            if (act_clipped) act_clipped[at] = has_box ? fminf(fmaxf(act, alow[ai]), ahigh[ai]) : act;
          }
        }
        u[i] = (lp + 0.f) + (0.f + 0.f);                 // wave_sum_fast over lane = action: lanes 16.. hold 0
        ue[i] = (entl + 0.f) + (0.f + 0.f);
      }
      const float lp_row = xor32_sum(xor16_sum((u[0] + u[1]) + (u[2] + u[3])));
      const float ent_row = xor32_sum(xor16_sum((ue[0] + ue[1]) + (ue[2] + ue[3])));
      if (live && q4 == 0) {
        if (log_prob) log_prob[n] = lp_row;
        if (entropy) entropy[n] = ent_row;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 wo = *reinterpret_cast<const f32x4*>(cw + CST_WO + 16 * t);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[t][i] = 16 * t + 4 * q4 + i < H2 ? wo[i] * z[t][i] : 0.f;
      }
      const float v = tile_sum64(z) + cst[w][CST_BO];
      float* const dst = w == 1 ? v_r : v_c;
      if (live && q4 == 0 && dst) dst[n] = v;
    }
  }
}

template <int CIT>
__global__ void __launch_bounds__(64) cost_forward_kernel(icrl_costnet_t cn, CnLayout cl, const double* obs,
                                                          const float* acs, int N, float* cost, int mode = 0) {
  __shared__ __attribute__((aligned(16))) float cx[MAX_CN_IN];
  __shared__ __attribute__((aligned(16))) float ch[2][MAX_H];
  CnRegs<CIT> C;
  load_cn_regs<CIT>(cn, cl, C);
  const int n = blockIdx.x;
  const int AS = cn.is_discrete ? 1 : cn.acs_dim;
  const float c = cost_forward_wave<CIT>(cn, cl, C, obs + (size_t)n * cn.obs_dim, acs + (size_t)n * AS, cx, ch, mode);
  if (threadIdx.x == 0) cost[n] = c;
}

__global__ void __launch_bounds__(64) env_step_kernel(icrl_env_t e, const float* actions, double* raw_rew, uint8_t* dones) {
  __shared__ double s_old[MAX_OBS];
  const int n = blockIdx.x;
  for (int i = threadIdx.x; i < e.obs_dim; i += WAVE) s_old[i] = e.s[(size_t)n * e.obs_dim + i];
  __syncthreads();
  double rew; int done;
  uint32_t ctr = e.step_count[n];
  int tep = e.t_ep[n];
  env_step_wave(e, n, s_old, actions + (size_t)n * e.act_dim, e.key[n], ctr, tep, nullptr, rew, done);
  if (threadIdx.x == 0) { raw_rew[n] = rew; dones[n] = (uint8_t)done; }
}

__global__ void env_reset_kernel(icrl_env_t e) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int O = e.obs_dim;
  if (idx < e.n_envs * O) {
    const int n = idx / O, i = idx % O;
    e.s[idx] = env_reset_value(e, e.key[n], e.step_count[n], i);
    if (i == 0) e.t_ep[n] = 0;
  }
}

// =================================================================================================================
// persistent single-env episode sampler: sample_from_agent / evaluate_policy (icrl/utils.py:323-357, evaluation.py:10-67)
// One workgroup per independent env stream runs its episodes start to finish inside ONE launch: frozen-statistics
// normalise -> policy forward (sample or mode) -> clip -> env step (+auto-reset) -> record (s_{t+1}, a_t), reward.
// =================================================================================================================
struct SampleArgs {
  icrl_env_t env;          // arrays sized [n_streams]; env.s etc. are this call's scratch state
  icrl_norm_t nm;          // training must be 0: statistics are only read
  PolLayout pl;
  const float* PT;
  int pt_in_lds;
  const float* noise;      // [n_streams][rows_per_stream][act] or NULL (deterministic)
  const float* alow;
  const float* ahigh;
  int episodes_per_stream, rows_per_stream, deterministic, do_reset;
  const int* stream_row0;  // [n_streams] first row (noise in, outputs out) of every stream, or NULL: stream * rows_per_stream
  int total_rows;          // rows of the noise / output arrays (only read with stream_row0)
  double* orig_obs;        // [n_streams*rows_per_stream, obs] raw observation AFTER each step
  double* obs;             // same, normalised
  float* actions;          // [.., act_store] clipped action that produced it
  double* ep_rewards;      // [n_streams*episodes_per_stream]
  int* ep_lengths;         // [n_streams*episodes_per_stream]
};

template <int OCT>
__device__ __forceinline__ void sample_episodes_body(const SampleArgs& a) {
  __shared__ ActShared sh;
  PolRegs<OCT> R;
  load_pol_regs<OCT>(a.pl, a.PT, R);    // once for every step of every episode of this stream
  __shared__ int s_done;
  __shared__ double s_rew;
  __shared__ double Bl[MAX_OBS * MAX_ACT];     // dynamics matrix: read every step, kept out of the global-memory latency
  __shared__ float noise_s[MAX_ACT], alow_s[MAX_ACT], ahigh_s[MAX_ACT];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = a.pl.O, A = a.pl.A;
  const int AS = a.pl.discrete ? 1 : A;
  icrl_env_t env = a.env; globalize(env);
  // what the step loop reads of the argument block, once (a batched launch keeps the block in LDS; pointers: common.h as_global)
  const float* const noise_g = as_global(a.noise);
  float* const actions_g = as_global(a.actions);
  double* const orig_obs_g = as_global(a.orig_obs);
  double* const obs_g = as_global(a.obs);
  const int norm_obs = a.nm.norm_obs, deterministic = a.deterministic;
  const double clip_obs = a.nm.clip_obs;
  for (int i = tid; i < O * a.env.act_dim; i += 192) Bl[i] = a.env.B[i];
  env.B = Bl;
  const bool has_box = a.alow != nullptr && a.ahigh != nullptr;
  if (tid < MAX_ACT) { alow_s[tid] = (has_box && tid < A) ? a.alow[tid] : 0.f; ahigh_s[tid] = (has_box && tid < A) ? a.ahigh[tid] : 0.f; }
  const uint32_t e_key = a.env.key[n];
  uint32_t e_ctr = a.env.step_count[n];
  int e_tep = a.env.t_ep[n];
  // frozen normaliser statistics of this thread's observation components (i = tid, O <= 128 < 192)
  double n_mean = 0.0, n_den = 1.0;
  if (tid < O && a.nm.norm_obs) { n_mean = a.nm.obs_mean[tid]; n_den = sqrt(a.nm.obs_var[tid] + a.nm.epsilon); }
  for (int i = tid; i < MAX_OBS; i += 192) sh.x[i] = 0.f;
  if (tid < O) {
    double v = a.env.s[(size_t)n * O + tid];
    if (a.do_reset) v = env_reset_value(a.env, e_key, e_ctr, tid);
    sh.s_new[tid] = v;
  }
  if (a.do_reset) e_tep = 0;
  __syncthreads();
  size_t row = a.stream_row0 != nullptr ? (size_t)a.stream_row0[n] : (size_t)n * a.rows_per_stream;
  const size_t row_end = a.stream_row0 != nullptr ? (size_t)a.total_rows : row + a.rows_per_stream;
  float noise_reg = (noise_g != nullptr && tid < AS) ? noise_g[row * AS + tid] : 0.f;   // noise of the first step
  for (int ep = 0; ep < a.episodes_per_stream; ++ep) {
    double ep_rew = 0.0;
    int ep_len = 0;
    while (true) {
      if (tid < O) {
        const double raw = sh.s_new[tid];
        double o = raw;
        if (norm_obs) o = fmin(fmax((raw - n_mean) / n_den, -clip_obs), clip_obs);
        sh.s_old[tid] = raw;
        sh.x[tid] = (float)o;
      }
      if (noise_g != nullptr && tid < AS) {
        noise_s[tid] = noise_reg;
        if (row + 1 < row_end) noise_reg = noise_g[(row + 1) * AS + tid];      // next step's noise lands during this step
      }
      __syncthreads();
      policy_forward_block<OCT>(a.pl, R, sh, noise_g ? noise_s : nullptr, deterministic || noise_g == nullptr,
                                has_box ? alow_s : nullptr, has_box ? ahigh_s : nullptr);
      __syncthreads();
      const bool in_rows = row < row_end;      // (a stream whose speculative start row was too late can run off the arrays)
      if (w == 0) {
        double rew; int done;
        env_step_wave(env, n, sh.s_old, sh.act_clip, e_key, e_ctr, e_tep, sh.s_new, rew, done);
        if (lane == 0) { s_done = done; s_rew = rew; }
        if (lane < AS && in_rows) actions_g[row * AS + lane] = sh.act_clip[lane];
      }
      __syncthreads();
      if (tid < O && in_rows) {
        const double raw = sh.s_new[tid];
        double o = raw;
        if (norm_obs) o = fmin(fmax((raw - n_mean) / n_den, -clip_obs), clip_obs);
        orig_obs_g[row * O + tid] = raw;
        obs_g[row * O + tid] = o;
      }
      ep_rew += s_rew;     // episode_reward += reward (un-normalised: norm_reward is False on sampling / eval envs)
      ++ep_len;
      ++row;
      const int done = s_done;
      __syncthreads();
      if (done) break;
    }
    if (tid == 0) { a.ep_rewards[(size_t)n * a.episodes_per_stream + ep] = ep_rew; a.ep_lengths[(size_t)n * a.episodes_per_stream + ep] = ep_len; }
  }
}

template <int OCT>
__global__ void __launch_bounds__(192) sample_episodes_kernel(SampleArgs a) {
  sample_episodes_body<OCT>(a);
}

// several runs' streams in ONE launch: grid (n_streams, n_runs), run = blockIdx.y (streams of different runs never interact)
template <int OCT>
__global__ void __launch_bounds__(192) sample_episodes_batch_kernel(const SampleArgs* __restrict__ runs) {
  __shared__ SampleArgs a;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(runs + blockIdx.y);
    unsigned* dst = reinterpret_cast<unsigned*>(&a);
    for (unsigned i = threadIdx.x; i < sizeof(SampleArgs) / 4; i += 192) dst[i] = src[i];
  }
  __syncthreads();
  sample_episodes_body<OCT>(a);
}

// rows from which icrl_policy_forward / icrl_policy_evaluate take the 16-rows-per-pass MFMA kernel (below: one workgroup per row)
constexpr int ROWS_KERNEL_MIN = 64;

static bool dims_ok(const icrl_policy_t* p) {      // what the fused kernels serve: two hidden layers per branch, no shared trunk
  return p->arch == nullptr && p->obs_dim > 0 && p->obs_dim <= MAX_OBS && p->act_dim > 0 && p->act_dim <= MAX_ACT && p->h1 > 0 && p->h1 <= MAX_H &&
         p->h2 > 0 && p->h2 <= MAX_H;
}
static bool cn_ok(const icrl_costnet_t* cn) {
  return cn->in_dim > 0 && cn->in_dim <= MAX_CN_IN && cn->h1 > 0 && cn->h1 <= MAX_H && (cn->n_hidden == 1 ||
         (cn->n_hidden == 2 && cn->h2 > 0 && cn->h2 <= MAX_H)) && cn->obs_dim <= MAX_OBS && cn->acs_dim <= MAX_ACT;
}
static int bad_dims(const char* who, const icrl_policy_t* p) {
  return fail("%s: policy obs_dim %d (1..%d), act_dim %d (1..%d), hidden (%d, %d) (1..%d each)%s; wider policies, shared trunks and other depths run "
              "through the generic-shape entry points: icrl_policy_forward / icrl_policy_evaluate / icrl_ppo_lag_train and the per-step rollout", who,
              p->obs_dim, MAX_OBS, p->act_dim, MAX_ACT, p->h1, p->h2, MAX_H, p->arch != nullptr ? ", an `arch` descriptor" : "");
}
static int bad_cn(const char* who, const icrl_costnet_t* cn) {
  return fail("%s: constraint net in_dim %d (1..%d), %d hidden layers (1 or 2) of (%d, %d) (1..%d; wider and deeper nets: icrl_cost_mlp_forward / icrl_disc_reward / icrl_cn_train and the per-step rollout), obs_dim %d (<= %d), acs_dim %d (<= %d)", who,
              cn->in_dim, MAX_CN_IN, cn->n_hidden, cn->h1, cn->h2, MAX_H, cn->obs_dim, MAX_OBS, cn->acs_dim, MAX_ACT);
}

}  // namespace icrl

using namespace icrl;

extern "C" int icrl_policy_prepare(const icrl_policy_t* p, void* stream) {
  if (policy_is_wide(p)) return launch_generic_transpose(p, (hipStream_t)stream);      // generic-shape path: per-layer transposes (generic.hip)
  if (!dims_ok(p)) return bad_dims("icrl_policy_prepare", p);
  PolLayout L = make_pol_layout(p->obs_dim, p->act_dim, p->h1, p->h2, p->discrete);
  if (L.n != p->n_params) return fail("icrl_policy_prepare: n_params = %d, the layout needs %d", p->n_params, L.n);
  hipLaunchKernelGGL(policy_transpose_kernel, dim3((L.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, L, p->params, p->params_t);
  return (int)hipGetLastError();
}

extern "C" int icrl_costnet_prepare(const icrl_costnet_t* cn, void* stream) {
  if (cn->n_hidden > 2 || cn->n_hidden == 0) return 0;      // served 64 rows per workgroup from `params` (cn_train.hip): no transposed copy
  if (!costnet_is_wide(cn) && !cn_ok(cn)) return bad_cn("icrl_costnet_prepare", cn);
  CnLayout L = make_cn_layout(cn->in_dim, cn->n_hidden, cn->h1, cn->h2);
  if (L.n != cn->n_params) return fail("icrl_costnet_prepare: n_params = %d, the layout needs %d", cn->n_params, L.n);
  hipLaunchKernelGGL(costnet_transpose_kernel, dim3((L.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, L, cn->params, cn->params_t);
  return (int)hipGetLastError();
}

extern "C" int icrl_policy_forward(const icrl_policy_t* p, const double* obs, const float* noise, int N, int deterministic,
                                   const float* action_low, const float* action_high, float* actions, float* act_clipped,
                                   float* v_r, float* v_c, float* log_prob, void* stream) {
  if (N <= 0) return fail("policy forward / evaluate: N = %d rows", N);
  if (policy_is_wide(p))      // hidden widths above 64: the generic-shape kernel (generic.hip)
    return launch_policy_generic(p, obs, noise, N, deterministic, action_low, action_high, actions, act_clipped, v_r, v_c, log_prob, nullptr, nullptr,
                                 (hipStream_t)stream);
  if (!dims_ok(p)) return bad_dims("policy forward / evaluate", p);
  PolLayout L = make_pol_layout(p->obs_dim, p->act_dim, p->h1, p->h2, p->discrete);
  if (N >= ROWS_KERNEL_MIN && !p->discrete) {
    const int grid = (N + 15) / 16 < 1024 ? (N + 15) / 16 : 1024;
    if (L.O <= 32)
      hipLaunchKernelGGL(policy_rows_kernel<2>, dim3(grid), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs, noise, deterministic,
                         action_low, action_high, actions, act_clipped, v_r, v_c, log_prob, (const float*)nullptr, (float*)nullptr, N);
    else
      hipLaunchKernelGGL(policy_rows_kernel<8>, dim3(grid), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs, noise, deterministic,
                         action_low, action_high, actions, act_clipped, v_r, v_c, log_prob, (const float*)nullptr, (float*)nullptr, N);
    return (int)hipGetLastError();
  }
  if (L.O <= 32)
    hipLaunchKernelGGL(policy_forward_kernel<2>, dim3(N), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs, noise,
                       deterministic, action_low, action_high, actions, act_clipped, v_r, v_c, log_prob,
                       (const float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL(policy_forward_kernel<8>, dim3(N), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs, noise,
                       deterministic, action_low, action_high, actions, act_clipped, v_r, v_c, log_prob,
                       (const float*)nullptr, (float*)nullptr);
  return (int)hipGetLastError();
}

extern "C" int icrl_policy_evaluate(const icrl_policy_t* p, const double* obs, const float* actions, int N, float* v_r,
                                    float* v_c, float* log_prob, float* entropy, void* stream) {
  if (N <= 0) return fail("policy forward / evaluate: N = %d rows", N);
  if (policy_is_wide(p))
    return launch_policy_generic(p, obs, nullptr, N, 1, nullptr, nullptr, nullptr, nullptr, v_r, v_c, log_prob, actions, entropy, (hipStream_t)stream);
  if (!dims_ok(p)) return bad_dims("policy forward / evaluate", p);
  PolLayout L = make_pol_layout(p->obs_dim, p->act_dim, p->h1, p->h2, p->discrete);
  if (N >= ROWS_KERNEL_MIN && !p->discrete) {
    const int grid = (N + 15) / 16 < 1024 ? (N + 15) / 16 : 1024;
    if (L.O <= 32)
      hipLaunchKernelGGL(policy_rows_kernel<2>, dim3(grid), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs, (const float*)nullptr, 1,
                         (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, v_r, v_c, log_prob, actions, entropy, N);
    else
      hipLaunchKernelGGL(policy_rows_kernel<8>, dim3(grid), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs, (const float*)nullptr, 1,
                         (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, v_r, v_c, log_prob, actions, entropy, N);
    return (int)hipGetLastError();
  }
  if (L.O <= 32)
    hipLaunchKernelGGL(policy_forward_kernel<2>, dim3(N), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs,
                       (const float*)nullptr, 1, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                       v_r, v_c, log_prob, actions, entropy);
  else
    hipLaunchKernelGGL(policy_forward_kernel<8>, dim3(N), dim3(192), 0, (hipStream_t)stream, L, p->params_t, obs,
                       (const float*)nullptr, 1, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                       v_r, v_c, log_prob, actions, entropy);
  return (int)hipGetLastError();
}

// sample_episodes_kernel for the policies of the generic-shape path (layers above 64 units, shared trunk, other depths): the same
// persistent episode loop, one workgroup per stream, with the table-driven forward of generic.h on the trunk + policy branch (slot 0;
// the value branches are not needed here) and the weights read from the per-layer transposes every step (L2-resident).
struct GenSampleArgs {
  icrl_env_t env;
  icrl_norm_t nm;
  const float* P;
  const float* PT;
  const float* noise;
  const float* alow;
  const float* ahigh;
  int episodes_per_stream, rows_per_stream, deterministic, do_reset;
  const int* stream_row0;
  int total_rows;
  double* orig_obs;
  double* obs;
  float* actions;
  double* ep_rewards;
  int* ep_lengths;
};

__global__ void __launch_bounds__(GEN_MAX_H) sample_episodes_generic_kernel(GenNet net, GenSampleArgs a) {
  __shared__ float x[MAX_OBS];
  __shared__ float act[GEN_MAX_ROW];
  __shared__ double s_old[MAX_OBS], s_new[MAX_OBS], n_mean[MAX_OBS], n_den[MAX_OBS];
  __shared__ double Bl[MAX_OBS * MAX_ACT];
  __shared__ float act_raw[MAX_ACT], act_clip[MAX_ACT], noise_s[MAX_ACT], alow_s[MAX_ACT], ahigh_s[MAX_ACT];
  __shared__ int s_done;
  __shared__ double s_rew;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int O = net.O, A = net.A, AS = net.discrete ? 1 : A;
  icrl_env_t env = a.env;
  const int norm_obs = a.nm.norm_obs, deterministic = a.deterministic;
  const double clip_obs = a.nm.clip_obs;
  for (int i = tid; i < O * a.env.act_dim; i += nt) Bl[i] = a.env.B[i];
  env.B = Bl;
  const bool has_box = a.alow != nullptr && a.ahigh != nullptr;
  if (tid < MAX_ACT) { alow_s[tid] = (has_box && tid < A) ? a.alow[tid] : 0.f; ahigh_s[tid] = (has_box && tid < A) ? a.ahigh[tid] : 0.f; }
  const uint32_t e_key = a.env.key[n];
  uint32_t e_ctr = a.env.step_count[n];
  int e_tep = a.env.t_ep[n];
  for (int i = tid; i < O; i += nt) {      // frozen normaliser statistics; the state the stream starts from
    n_mean[i] = a.nm.norm_obs ? a.nm.obs_mean[i] : 0.0;
    n_den[i] = a.nm.norm_obs ? sqrt(a.nm.obs_var[i] + a.nm.epsilon) : 1.0;
    double v = a.env.s[(size_t)n * O + i];
    if (a.do_reset) v = env_reset_value(a.env, e_key, e_ctr, i);
    s_new[i] = v;
  }
  if (a.do_reset) e_tep = 0;
  __syncthreads();
  size_t row = a.stream_row0 != nullptr ? (size_t)a.stream_row0[n] : (size_t)n * a.rows_per_stream;
  const size_t row_end = a.stream_row0 != nullptr ? (size_t)a.total_rows : row + a.rows_per_stream;
  for (int ep = 0; ep < a.episodes_per_stream; ++ep) {
    double ep_rew = 0.0;
    int ep_len = 0;
    while (true) {
      for (int i = tid; i < O; i += nt) {
        const double raw = s_new[i];
        double o = raw;
        if (norm_obs) o = fmin(fmax((raw - n_mean[i]) / n_den[i], -clip_obs), clip_obs);
        s_old[i] = raw;
        x[i] = (float)o;
      }
      const bool in_rows = row < row_end;      // (a stream whose speculative start row was too late can run off the arrays)
      if (a.noise != nullptr && tid < AS) noise_s[tid] = in_rows ? a.noise[row * AS + tid] : 0.f;
      __syncthreads();
      gen_mlp_forward(net, a.PT, x, act, 0, tid);
      if (tid == 0) {
        float lp, ent;
        gen_policy_head(net, a.P, act + net.layer[net.head[0]].act_off, a.noise != nullptr ? noise_s : nullptr, deterministic || a.noise == nullptr,
                        has_box ? alow_s : nullptr, has_box ? ahigh_s : nullptr, nullptr, act_raw, act_clip, lp, ent);
      }
      __syncthreads();
      if (w == 0) {
        double rew; int done;
        env_step_wave(env, n, s_old, act_clip, e_key, e_ctr, e_tep, s_new, rew, done);
        if (lane == 0) { s_done = done; s_rew = rew; }
        if (lane < AS && in_rows) a.actions[row * AS + lane] = act_clip[lane];
      }
      __syncthreads();
      if (in_rows)
        for (int i = tid; i < O; i += nt) {
          const double raw = s_new[i];
          double o = raw;
          if (norm_obs) o = fmin(fmax((raw - n_mean[i]) / n_den[i], -clip_obs), clip_obs);
          a.orig_obs[row * O + i] = raw;
          a.obs[row * O + i] = o;
        }
      ep_rew += s_rew;     // episode_reward += reward (un-normalised: norm_reward is False on sampling / eval envs)
      ++ep_len;
      ++row;
      const int done = s_done;
      __syncthreads();
      if (done) break;
    }
    if (tid == 0) { a.ep_rewards[(size_t)n * a.episodes_per_stream + ep] = ep_rew; a.ep_lengths[(size_t)n * a.episodes_per_stream + ep] = ep_len; }
  }
}

static int make_sample_args(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol, const float* noise,
                            const float* action_low, const float* action_high, int episodes_per_stream, int rows_per_stream,
                            int deterministic, int do_reset, const int32_t* stream_row0, int total_rows, double* orig_obs, double* obs,
                            float* actions, double* ep_rewards, int32_t* ep_lengths, SampleArgs& a) {
  if (!dims_ok(pol)) return bad_dims("icrl_sample_episodes", pol);
  if (env->obs_dim != pol->obs_dim || nm->training)
    return fail("icrl_sample_episodes: env obs_dim %d vs policy %d; the normaliser must be frozen (training = %d)", env->obs_dim, pol->obs_dim, nm->training);
  if (episodes_per_stream * env->max_steps > rows_per_stream)
    return fail("icrl_sample_episodes: %d episodes x %d steps do not fit %d rows per stream", episodes_per_stream, env->max_steps, rows_per_stream);
  if (stream_row0 != nullptr && total_rows < 1) return fail("icrl_sample_episodes: stream_row0 given with total_rows = %d", total_rows);
  a.stream_row0 = stream_row0; a.total_rows = total_rows;
  a.env = *env; a.nm = *nm; a.pl = make_pol_layout(pol->obs_dim, pol->act_dim, pol->h1, pol->h2, pol->discrete);
  a.PT = pol->params_t; a.noise = noise; a.alow = action_low; a.ahigh = action_high;
  a.episodes_per_stream = episodes_per_stream; a.rows_per_stream = rows_per_stream; a.deterministic = deterministic;
  a.do_reset = do_reset; a.orig_obs = orig_obs; a.obs = obs; a.actions = actions; a.ep_rewards = ep_rewards; a.ep_lengths = ep_lengths;
  a.pt_in_lds = 0;
  return 0;
}

extern "C" int icrl_sample_episodes(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol, const float* noise,
                                    const float* action_low, const float* action_high, int episodes_per_stream,
                                    int rows_per_stream, int deterministic, int do_reset, const int32_t* stream_row0, int total_rows,
                                    double* orig_obs, double* obs, float* actions, double* ep_rewards, int32_t* ep_lengths,
                                    void* stream) {
  if (policy_is_wide(pol)) {      // generic-shape path: the same persistent loop with the table-driven forward
    GenNet net;
    if (int e = make_gen_net(pol, &net, "icrl_sample_episodes")) return e;
    if (env->obs_dim != pol->obs_dim || nm->training || env->obs_dim > MAX_OBS || env->act_dim > MAX_ACT || pol->params_t == nullptr)
      return fail("icrl_sample_episodes (generic-shape path): env obs_dim %d vs policy %d (<= %d), act_dim %d (<= %d); the normaliser must be frozen "
                  "(training = %d); params_t %s", env->obs_dim, pol->obs_dim, MAX_OBS, env->act_dim, MAX_ACT, nm->training, pol->params_t ? "set" : "NULL");
    if (episodes_per_stream * env->max_steps > rows_per_stream)
      return fail("icrl_sample_episodes: %d episodes x %d steps do not fit %d rows per stream", episodes_per_stream, env->max_steps, rows_per_stream);
    if (stream_row0 != nullptr && total_rows < 1) return fail("icrl_sample_episodes: stream_row0 given with total_rows = %d", total_rows);
    GenSampleArgs g{*env, *nm, pol->params, pol->params_t, noise, action_low, action_high, episodes_per_stream, rows_per_stream, deterministic, do_reset,
                    stream_row0, total_rows, orig_obs, obs, actions, ep_rewards, ep_lengths};
    hipLaunchKernelGGL(sample_episodes_generic_kernel, dim3(env->n_envs), dim3(net.W), 0, (hipStream_t)stream, net, g);
    return (int)hipGetLastError();
  }
  SampleArgs a;
  const int bad = make_sample_args(env, nm, pol, noise, action_low, action_high, episodes_per_stream, rows_per_stream, deterministic,
                                   do_reset, stream_row0, total_rows, orig_obs, obs, actions, ep_rewards, ep_lengths, a);
  if (bad) return bad;
  if (a.pl.O <= 32) hipLaunchKernelGGL(sample_episodes_kernel<2>, dim3(env->n_envs), dim3(192), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(sample_episodes_kernel<8>, dim3(env->n_envs), dim3(192), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int icrl_sample_episodes_batch(int n_runs, const icrl_sample_job_t* jobs, const float* action_low, const float* action_high,
                                          int episodes_per_stream, int rows_per_stream, int deterministic, int do_reset,
                                          void* args_ws, long long args_ws_bytes, void* stream) {
  static_assert(sizeof(SampleArgs) <= ICRL_BATCH_ARGS_BYTES, "ICRL_BATCH_ARGS_BYTES");
  if (n_runs < 1 || n_runs > 65535) return fail("icrl_sample_episodes_batch: n_runs = %d (1..65535)", n_runs);
  if (args_ws == nullptr || args_ws_bytes < (long long)n_runs * ICRL_BATCH_ARGS_BYTES)
    return fail("icrl_sample_episodes_batch: args_ws holds %lld B, %d runs need %lld", args_ws_bytes, n_runs, (long long)n_runs * ICRL_BATCH_ARGS_BYTES);
  hipStream_t s = (hipStream_t)stream;
  SampleArgs* d_args = (SampleArgs*)args_ws;
  const icrl_sample_job_t& j0 = jobs[0];
  for (int r = 0; r < n_runs; ++r) {
    const icrl_sample_job_t& j = jobs[r];
    if (j.env->n_envs != j0.env->n_envs || j.env->obs_dim != j0.env->obs_dim || j.pol->act_dim != j0.pol->act_dim ||
        j.pol->discrete != j0.pol->discrete || j.env->max_steps != j0.env->max_steps)
      return fail("icrl_sample_episodes_batch: run %d differs from run 0 in a shape (streams / obs / act / discrete / max_steps)", r);
    SampleArgs a;
    const int bad = make_sample_args(j.env, j.nm, j.pol, j.noise, action_low, action_high, episodes_per_stream, rows_per_stream,
                                     deterministic, do_reset, j.stream_row0, j.total_rows, j.orig_obs, j.obs, j.actions, j.ep_rewards,
                                     j.ep_lengths, a);
    if (bad) return bad;
    const int e = put_args(a, d_args + r, s);
    if (e) return e;
  }
  if (j0.pol->obs_dim <= 32) hipLaunchKernelGGL(sample_episodes_batch_kernel<2>, dim3(j0.env->n_envs, n_runs), dim3(192), 0, s, d_args);
  else hipLaunchKernelGGL(sample_episodes_batch_kernel<8>, dim3(j0.env->n_envs, n_runs), dim3(192), 0, s, d_args);
  return (int)hipGetLastError();
}

extern "C" int icrl_cost_mlp_forward(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* cost,
                                     void* stream) {
  if (N <= 0) return fail("cost forward: N = %d rows", N);
  if (costnet_is_wide(cn)) return launch_cn_cost_rows(cn, obs, acs, N, cost, 0, (hipStream_t)stream);
  if (!cn_ok(cn)) return bad_cn("cost forward", cn);
  CnLayout L = make_cn_layout(cn->in_dim, cn->n_hidden, cn->h1, cn->h2);
  if (cn->in_dim <= 32) hipLaunchKernelGGL(cost_forward_kernel<2>, dim3(N), dim3(64), 0, (hipStream_t)stream, *cn, L, obs, acs, N, cost, 0);
  else hipLaunchKernelGGL(cost_forward_kernel<10>, dim3(N), dim3(64), 0, (hipStream_t)stream, *cn, L, obs, acs, N, cost, 0);
  return (int)hipGetLastError();
}

extern "C" int icrl_disc_reward(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* out, int apply_log,
                                void* stream) {
  if (N <= 0) return fail("cost forward: N = %d rows", N);
  if (costnet_is_wide(cn)) return launch_cn_cost_rows(cn, obs, acs, N, out, apply_log ? 2 : 1, (hipStream_t)stream);
  if (!cn_ok(cn)) return bad_cn("cost forward", cn);
  CnLayout L = make_cn_layout(cn->in_dim, cn->n_hidden, cn->h1, cn->h2);
  const int mode = apply_log ? 2 : 1;
  if (cn->in_dim <= 32) hipLaunchKernelGGL(cost_forward_kernel<2>, dim3(N), dim3(64), 0, (hipStream_t)stream, *cn, L, obs, acs, N, out, mode);
  else hipLaunchKernelGGL(cost_forward_kernel<10>, dim3(N), dim3(64), 0, (hipStream_t)stream, *cn, L, obs, acs, N, out, mode);
  return (int)hipGetLastError();
}

extern "C" int icrl_synth_env_reset(const icrl_env_t* env, void* stream) {
  if (env->obs_dim > MAX_OBS || env->act_dim > MAX_ACT) return fail("synthetic env: obs_dim %d (<= %d), act_dim %d (<= %d)", env->obs_dim, MAX_OBS, env->act_dim, MAX_ACT);
  const int total = env->n_envs * env->obs_dim;
  hipLaunchKernelGGL(env_reset_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *env);
  return (int)hipGetLastError();
}

extern "C" int icrl_synth_env_step(const icrl_env_t* env, const float* actions, double* raw_rew, uint8_t* dones, void* stream) {
  if (env->obs_dim > MAX_OBS || env->act_dim > MAX_ACT) return fail("synthetic env: obs_dim %d (<= %d), act_dim %d (<= %d)", env->obs_dim, MAX_OBS, env->act_dim, MAX_ACT);
  hipLaunchKernelGGL(env_step_kernel, dim3(env->n_envs), dim3(64), 0, (hipStream_t)stream, *env, actions, raw_rew, dones);
  return (int)hipGetLastError();
}

static void launch_norm_step(const NormStepArgs& a, hipStream_t s) {
  if (a.N <= 128 && a.N * a.O <= NORM_CHUNK) hipLaunchKernelGGL(norm_step_small_kernel, dim3(1), dim3(1024), 0, s, a);
  else hipLaunchKernelGGL(norm_step_kernel, dim3(1), dim3(1024), 0, s, a);
}

extern "C" int icrl_vecnorm_reset(const icrl_norm_t* nm, const double* raw_obs, int N, int obs_dim, double* obs_out, void* stream) {
  if (N <= 0 || N > NORM_MAX_N || obs_dim > MAX_OBS)
    return fail("VecNormalize step: %d envs (1..%d per GPU: the float64 statistics are one numpy-ordered chain per column), obs_dim %d (<= %d)", N, NORM_MAX_N, obs_dim, MAX_OBS);
  hipLaunchKernelGGL(norm_reset_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *nm, raw_obs, N, obs_dim, obs_out);
  return (int)hipGetLastError();
}

extern "C" int icrl_vecnorm_step(const icrl_norm_t* nm, const double* raw_obs, const double* raw_rew, const float* raw_cost,
                                 const uint8_t* dones, int N, int obs_dim, double* obs_out, double* rew_out, double* cost_out,
                                 void* stream) {
  if (N <= 0 || N > NORM_MAX_N || obs_dim > MAX_OBS)
    return fail("VecNormalize step: %d envs (1..%d per GPU: the float64 statistics are one numpy-ordered chain per column), obs_dim %d (<= %d)", N, NORM_MAX_N, obs_dim, MAX_OBS);
  NormStepArgs a{*nm, raw_obs, raw_rew, raw_cost, dones, N, obs_dim, obs_out, rew_out, cost_out, nullptr, nullptr, nullptr, nullptr};
  launch_norm_step(a, (hipStream_t)stream);
  return (int)hipGetLastError();
}

extern "C" int icrl_debug_rollout_profile_wide(unsigned long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rollout_prof_wide), sizeof(unsigned long long) * 16);
}
extern "C" int icrl_debug_rollout_trace_wide(unsigned long long* out, int n_workgroups) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_trace), sizeof(unsigned long long) * 4 * (size_t)(n_workgroups < NORM_MAX_N_TRACE ? n_workgroups : NORM_MAX_N_TRACE));
}

extern "C" int icrl_debug_rollout_profile(unsigned long long* out4) {
  return (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_rollout_prof), sizeof(unsigned long long) * 8);
}

// all N workgroups of the persistent rollout must be co-resident (they wait for each other every step): ask the runtime
// how many blocks of this kernel one CU takes.  The answer is advisory (MI355X_MICROARCH.md: it can read one high for
// SGPR-heavy kernels), hence the bounded spins + status word as the backstop.
template <typename K>
static bool persistent_fits(K kernel, int blocks, size_t dyn_lds = 0) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (dyn_lds > 48 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, dyn_lds) != hipSuccess) return false;
  if (getenv("ICRL_DEBUG")) fprintf(stderr, "icrl: persistent kernel: %d workgroup(s) per CU x %d CUs (%zu B dynamic LDS), %d needed\n", per_cu, cus, dyn_lds, blocks);
  return (long long)per_cu * cus >= blocks;
}

// multi-env kernel: E envs per workgroup, G = ceil(N / E) workgroups per run, at most four statistics owners per workgroup
static bool multi_shape(int N, int n_stats, int* E, int* G) {
  const int gmin = (n_stats + 3) / 4;
  const int per = N / gmin;
  const char* force = getenv("ICRL_MULTI_E");      // tools only: force the envs per workgroup (4 | 8 | 16)
  if (force != nullptr && (atoi(force) == 4 || atoi(force) == 8 || atoi(force) == 16) && per >= atoi(force)) *E = atoi(force);
  else if (per >= 16 && N > 8 * 256) *E = 16;        // more than 2048 envs: 16 per workgroup keep the grid at <= 256 workgroups
  else if (per >= 8) *E = 8;
  else if (per >= 4) *E = 4;
  else return false;
  *G = (N + *E - 1) / *E;
  return *G >= gmin;
}

template <int OCT, int CIT, int E>
static int launch_multi_e(const WideArgs* one, const WideArgs* d_args, int n_runs, int G, size_t dyn, hipStream_t s) {
  if (one != nullptr) {
    if (!persistent_fits(rollout_multi_kernel<OCT, CIT, E>, G, dyn)) return -1;
    WideArgs arg = *one;
    static const bool no_pack = getenv("ICRL_NO_XCD_PACK") != nullptr;
    const int packed = !no_pack && arg.xcc != nullptr && G <= 32;
    void* params[] = {(void*)&arg, (void*)&packed};
    const hipError_t e = hipLaunchCooperativeKernel((const void*)rollout_multi_kernel<OCT, CIT, E>, dim3(packed ? 8 * (G - 1) + 1 : G), dim3(256), params, (unsigned)dyn, s);
    if (e == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return -1; }
    return (int)e;
  } else {
    if (!persistent_fits(rollout_multi_batch_kernel<OCT, CIT, E>, G, dyn)) return -1;
    // packed layout (a run's G workgroups on ONE XCD: workgroups b, b + 8, ...) when every XCD can hold its share of the grid at once
    // (32 CUs each); otherwise run-major, whose runs become resident oldest first
    int per_cu = 0;
    const int groups = (n_runs + 7) / 8;
    static const bool no_pack_b = getenv("ICRL_NO_XCD_PACK") != nullptr;
    const bool packed = !no_pack_b && G <= 32 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rollout_multi_batch_kernel<OCT, CIT, E>, 256, dyn) == hipSuccess &&
                        groups * G <= 32 * per_cu;
    if (packed) hipLaunchKernelGGL((rollout_multi_batch_kernel<OCT, CIT, E>), dim3(8 * G * ((n_runs + 7) / 8)), dim3(256), dyn, s, d_args, n_runs, G, 1);
    else hipLaunchKernelGGL((rollout_multi_batch_kernel<OCT, CIT, E>), dim3(G, n_runs), dim3(256), dyn, s, d_args, n_runs, G, 0);
  }
  return (int)hipGetLastError();
}

// one: single-run launch (argument block by value) | d_args: n_runs blocks in device memory.  -1: does not fit the device
static int launch_multi(bool small, bool cn128, int E, const WideArgs* one, const WideArgs* d_args, int n_runs, int G, size_t dyn, hipStream_t s) {
  // cn128: the cost net reads <= 128 inputs (AntWall: 121) — 32 instead of 40 first-layer k steps in the register image
  if (!small && cn128)
    return E == 16 ? launch_multi_e<8, 8, 16>(one, d_args, n_runs, G, dyn, s)
           : (E == 8 ? launch_multi_e<8, 8, 8>(one, d_args, n_runs, G, dyn, s) : launch_multi_e<8, 8, 4>(one, d_args, n_runs, G, dyn, s));
  if (small) return E == 16 ? launch_multi_e<2, 2, 16>(one, d_args, n_runs, G, dyn, s)
                    : (E == 8 ? launch_multi_e<2, 2, 8>(one, d_args, n_runs, G, dyn, s) : launch_multi_e<2, 2, 4>(one, d_args, n_runs, G, dyn, s));
  return E == 16 ? launch_multi_e<8, 10, 16>(one, d_args, n_runs, G, dyn, s)
         : (E == 8 ? launch_multi_e<8, 10, 8>(one, d_args, n_runs, G, dyn, s) : launch_multi_e<8, 10, 4>(one, d_args, n_runs, G, dyn, s));
}

extern "C" int icrl_rollout_collect_ex(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol,
                                       const icrl_costnet_t* cn, const icrl_buffer_t* buf, const icrl_agent_t* ag,
                                       const float* noise, const float* action_low, const float* action_high,
                                       double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                                       int do_gae, void* stream) {
  const int N = env->n_envs, O = env->obs_dim, T = buf->T;
  const bool generic = policy_is_wide(pol) || (cn != nullptr && costnet_is_wide(cn));      // shapes of the generic-shape path: per-step launches
  if (!generic && !dims_ok(pol)) return bad_dims("icrl_rollout_collect", pol);
  if (pol->obs_dim != O || buf->N != N || buf->obs_dim != O)
    return fail("icrl_rollout_collect: env (%d envs, obs_dim %d) vs policy obs_dim %d vs buffer (%d envs, obs_dim %d)", N, O, pol->obs_dim, buf->N, buf->obs_dim);
  if (!generic && cn != nullptr && !cn_ok(cn)) return bad_cn("icrl_rollout_collect", cn);
  if (N > NORM_MAX_N)
    return fail("icrl_rollout_collect: %d envs on one GPU, limit %d (shard the envs over ranks: the float64 normaliser statistics are one "
                "numpy-ordered chain per column)", N, NORM_MAX_N);
  hipStream_t s = (hipStream_t)stream;
  if (generic) {
    // a policy with layers above 64 units / an `arch` descriptor, or a constraint net above 64 units / two layers: the reference's
    // per-step loop as four launches per step on this stream — policy_generic_kernel (generic.hip), cn_cost_rows_kernel (cn_train.hip),
    // act_step_generic_kernel, the normaliser — no host work in between
    if (O > MAX_OBS || env->act_dim > MAX_ACT || T < 1)
      return fail("icrl_rollout_collect (generic-shape path): obs_dim %d (<= %d), act_dim %d (<= %d), T = %d", O, MAX_OBS, env->act_dim, MAX_ACT, T);
    if (buf->act_store != (pol->discrete ? 1 : pol->act_dim) || (!pol->discrete && env->act_dim != pol->act_dim))
      return fail("icrl_rollout_collect (generic-shape path): buffer act_store %d / env act_dim %d vs policy act_dim %d (discrete: act_store 1)",
                  buf->act_store, env->act_dim, pol->act_dim);
    // ONE persistent launch (rollout_generic_kernel: the loop of rollout_persistent_kernel around the table-driven forward) when the
    // shapes are those of the one-workgroup-per-env kernel and the constraint net fits its register image; do_gae & 2 forces the per-step launches
    if (policy_is_wide(pol) && (cn == nullptr || (!costnet_is_wide(cn) && cn_ok(cn))) && !(do_gae & 2) && N <= 128 && N * O <= NORM_CHUNK &&
        O * env->act_dim <= MAX_OBS * MAX_ACT && (size_t)N * (2 * O + 4) <= (size_t)256 * GRAN_MAX && pol->params_t != nullptr) {
      GenRolloutArgs ga;
      if (int e = make_gen_net(pol, &ga.net, "icrl_rollout_collect")) return e;
      ga.P = pol->params;
      const int G2 = 2 * O + 4;
      const size_t need = (size_t)16 * N * O + (size_t)16 * N + (size_t)8 * N + (size_t)8 * N + 1024 + (size_t)16 * N * G2;
      char* ws = (ag->xch_ws != nullptr && (size_t)ag->xch_ws_bytes >= need) ? reinterpret_cast<char*>(ag->xch_ws)
                 : ((size_t)T * N * sizeof(float) >= need ? reinterpret_cast<char*>(buf->reward_advantages) : nullptr);
      if (ws != nullptr) {
        PersistArgs& p = ga.p;
        ActStepArgs& a = p.act;
        a.env = *env; a.buf = *buf; a.ag = *ag;
        a.pl = make_pol_layout(pol->obs_dim, pol->act_dim, MAX_H, MAX_H, pol->discrete);      // (only obs / act / discrete are read)
        a.PT = pol->params_t; a.noise = noise; a.alow = action_low; a.ahigh = action_high;
        a.has_cn = cn != nullptr;
        if (cn) { a.cn = *cn; a.cl = make_cn_layout(cn->in_dim, cn->n_hidden, cn->h1, cn->h2); }
        p.nm = *nm; p.T = T; p.prof = (do_gae & 4) != 0;
        char* base = ws;
        p.xch_obs = reinterpret_cast<double*>(base); base += (size_t)16 * N * O;
        p.xch_rew = reinterpret_cast<double*>(base); base += (size_t)16 * N;
        p.xch_cost = reinterpret_cast<float*>(base); base += (size_t)8 * N;
        p.xch_done = reinterpret_cast<unsigned*>(base); base += ((size_t)8 * N + 255) / 256 * 256;
        p.counter = reinterpret_cast<unsigned*>(base); base += 512;
        p.xg = reinterpret_cast<unsigned long long*>(base);
        p.g_magic = (unsigned)((1ull << 32) / (unsigned long long)(G2 / 2));
        hipError_t e = hipMemsetAsync(p.counter, 0, 512 + (size_t)16 * N * G2, s);
        if (e != hipSuccess) return (int)e;
        const bool small = O <= 32 && (!cn || cn->in_dim <= 32);
        const size_t dyn = persist_dyn_lds(N, O, env->act_dim);
        const int threads = 256;
        auto go = [&](auto kernel) -> int {      // -1: the grid is not co-resident
          int dev = 0, cus = 0, per_cu = 0;
          if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
          if (dyn > 48 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return -1;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn) != hipSuccess || (long long)per_cu * cus < N) return -1;
          const hipError_t e_ = launch_coresident(kernel, dim3(N), dim3(threads), dyn, s, ga);
          if (e_ == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return -1; }
          return (int)e_;
        };
        const int perr = small ? go(rollout_generic_kernel<2, 2>) : go(rollout_generic_kernel<8, 10>);
        if (perr >= 0) {
          const int err = perr != 0 ? perr : (int)hipGetLastError();
          if (err || !(do_gae & 1)) return err;
          return icrl_gae_dual_ws(buf->rewards, buf->costs, buf->reward_values, buf->cost_values, buf->dones, ag->last_v_r, ag->last_v_c,
                                  ag->last_dones, buf->reward_advantages, buf->cost_advantages, buf->reward_returns, buf->cost_returns, T, N,
                                  reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0, buf->gae_ws, buf->gae_ws_bytes, stream);
        }
      }
    }
    GenStepArgs g;
    g.env = *env; g.buf = *buf; g.ag = *ag; g.has_cn = cn != nullptr;
    const int AS = buf->act_store;
    for (int t = 0; t < T; ++t) {
      const size_t row = (size_t)t * N;
      int err = policy_is_wide(pol)
          ? launch_policy_generic(pol, ag->last_obs, noise ? noise + row * AS : nullptr, N, 0, action_low, action_high, buf->actions + row * AS, ag->act_clipped,
                                  buf->reward_values + row, buf->cost_values + row, buf->log_probs + row, nullptr, nullptr, s)
          : icrl_policy_forward(pol, ag->last_obs, noise ? noise + row * AS : nullptr, N, 0, action_low, action_high, buf->actions + row * AS, ag->act_clipped,
                                buf->reward_values + row, buf->cost_values + row, buf->log_probs + row, stream);
      if (err) return err;
      if (cn != nullptr) {
        err = icrl_cost_mlp_forward(cn, env->s, ag->act_clipped, N, ag->raw_cost, stream);
        if (err) return err;
      }
      hipLaunchKernelGGL(act_step_generic_kernel, dim3(N), dim3(64), 0, s, g, t);
      NormStepArgs b{*nm, env->s, ag->raw_rew, cn ? ag->raw_cost : nullptr, ag->dones, N, O, ag->last_obs, nullptr, nullptr,
                     buf->new_observations + row * O, buf->rewards + row, buf->costs + row, ag->last_dones};
      launch_norm_step(b, s);
    }
    const int err = (int)hipGetLastError();
    if (err || !(do_gae & 1)) return err;
    return icrl_gae_dual_ws(buf->rewards, buf->costs, buf->reward_values, buf->cost_values, buf->dones, ag->last_v_r, ag->last_v_c,
                            ag->last_dones, buf->reward_advantages, buf->cost_advantages, buf->reward_returns, buf->cost_returns, T, N,
                            reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0, buf->gae_ws, buf->gae_ws_bytes, stream);
  }
  ActStepArgs a;
  a.env = *env; a.buf = *buf; a.ag = *ag;
  a.pl = make_pol_layout(pol->obs_dim, pol->act_dim, pol->h1, pol->h2, pol->discrete);
  a.PT = pol->params_t; a.noise = noise; a.alow = action_low; a.ahigh = action_high;
  a.has_cn = cn != nullptr;
  if (cn) { a.cn = *cn; a.cl = make_cn_layout(cn->in_dim, cn->n_hidden, cn->h1, cn->h2); }
  // several environments per workgroup, interleaved (rollout_multi_kernel): do_gae bit 5
  if (((do_gae & 32) || N > WIDE_MAX_N) && !(do_gae & 2) && nm->training && !pol->discrete && N <= NORM_MAX_N && T >= 1) {
    int E = 0, G = 0;
    const int n_stats = O + (cn ? 2 : 1);
    const size_t GX = 2 * (size_t)O + 4, GS = 4 * (size_t)O + 4;
    const size_t need = 16 * (size_t)N * GX + 16 * GS + 256;
    void* ws = (ag->xch_ws != nullptr && (size_t)ag->xch_ws_bytes >= need) ? ag->xch_ws
               : ((size_t)T * N * sizeof(float) >= need ? (void*)buf->reward_advantages : nullptr);
    if (multi_shape(N, n_stats, &E, &G) && ws != nullptr) {
      WideArgs p;
      p.act = a; p.nm = *nm; p.T = T; p.G = G; p.prof = (do_gae & 4) ? 1 : ((do_gae & 8) ? G : 0);
      p.xg = reinterpret_cast<unsigned long long*>(ws);
      p.sg = p.xg + 2 * (size_t)N * GX;
      p.xcc = G <= 32 ? p.sg + 2 * GS : nullptr;      // (the workgroups' XCD ids: the 256 spare bytes behind the statistics granules)
      hipError_t e = hipMemsetAsync(p.xg, 0, 16 * (size_t)N * GX + 16 * GS + (p.xcc ? 256 : 0), s);
      if (e != hipSuccess) return (int)e;
      const bool small = a.pl.O <= 32 && (!cn || cn->in_dim <= 32);
      const int err = launch_multi(small, !cn || cn->in_dim <= 128, E, &p, nullptr, 1, G, multi_dyn_lds(N, O, env->act_dim, (n_stats + G - 1) / G), s);
      if (err >= 0) {
        if (err || !(do_gae & 1)) return err;
        return icrl_gae_dual_ws(buf->rewards, buf->costs, buf->reward_values, buf->cost_values, buf->dones, ag->last_v_r,
                                ag->last_v_c, ag->last_dones, buf->reward_advantages, buf->cost_advantages, buf->reward_returns,
                                buf->cost_returns, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0, buf->gae_ws,
                                buf->gae_ws_bytes, stream);
      }
    }
  }
  // many environments: persistent launch with the statistics partitioned by observation column (rollout_wide_kernel); measured
  // against the replicated-statistics kernel at HC widths: 64 envs 9.6 vs 9.5 us per step, 128 envs 9.9 vs 13.9
  if (!(do_gae & 2) && nm->training && (N > 96 || N * O > NORM_CHUNK || (do_gae & 16)) && N <= WIDE_MAX_N && O * env->act_dim <= MAX_OBS * MAX_ACT && T >= 1) {
    const bool small = a.pl.O <= 32 && (!cn || cn->in_dim <= 32);
    const void* kfn = small ? (const void*)rollout_wide_kernel<2, 2> : (const void*)rollout_wide_kernel<8, 10>;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, 0) == hipSuccess && per_cu > 0) {
      const int max_g = per_cu * cus;
      const int E = (N + max_g - 1) / max_g;                 // envs per workgroup
      const int G = (N + E - 1) / E;                         // balanced grid
      const size_t GX = 2 * (size_t)O + 4, GS = 4 * (size_t)O + 4;
      const size_t need = 16 * (size_t)N * GX + 16 * GS + 256;
      void* ws = (ag->xch_ws != nullptr && (size_t)ag->xch_ws_bytes >= need) ? ag->xch_ws
                 : ((size_t)T * N * sizeof(float) >= need ? (void*)buf->reward_advantages : nullptr);
      if (getenv("ICRL_DEBUG")) fprintf(stderr, "icrl_rollout_collect: wide persistent kernel N=%d obs=%d: %d workgroup(s) per CU x %d CUs -> E=%d envs per workgroup, grid %d\n", N, O, per_cu, cus, E, G);
      if (E <= WIDE_E && G >= O + 2 && ws != nullptr) {
        WideArgs p;
        p.act = a; p.nm = *nm; p.T = T; p.G = G; p.prof = (do_gae & 4) ? 1 : ((do_gae & 8) ? G : 0);
        p.xg = reinterpret_cast<unsigned long long*>(ws);
        p.sg = p.xg + 2 * (size_t)N * GX;
        p.xcc = nullptr;
        hipError_t e = hipMemsetAsync(p.xg, 0, 16 * (size_t)N * GX + 16 * GS, s);
        if (e != hipSuccess) return (int)e;
        int err = p.prof ? (int)(small ? launch_coresident(rollout_wide_kernel<2, 2, true>, dim3(G), dim3(256), 0, s, p)
                                       : launch_coresident(rollout_wide_kernel<8, 10, true>, dim3(G), dim3(256), 0, s, p))
                         : (int)(small ? launch_coresident(rollout_wide_kernel<2, 2>, dim3(G), dim3(256), 0, s, p)
                                       : launch_coresident(rollout_wide_kernel<8, 10>, dim3(G), dim3(256), 0, s, p));
        if (err == (int)hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); goto per_step; }
        if (err || !(do_gae & 1)) return err;
        return icrl_gae_dual_ws(buf->rewards, buf->costs, buf->reward_values, buf->cost_values, buf->dones, ag->last_v_r,
                             ag->last_v_c, ag->last_dones, buf->reward_advantages, buf->cost_advantages, buf->reward_returns,
                             buf->cost_returns, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0, buf->gae_ws,
                          buf->gae_ws_bytes, stream);
      }
    }
  }
  // persistent path: one launch for all T steps (see rollout_persistent_kernel).  Its exchange arrays live in the not yet
  // computed reward_advantages plane of the buffer (GAE fills that afterwards).  do_gae & 2 forces the per-step launches.
  {
    const int G = 2 * O + 4;
    const bool gran = (size_t)N * G <= (size_t)256 * GRAN_MAX;
    const size_t need = (size_t)16 * N * O + (size_t)16 * N + (size_t)8 * N + (size_t)8 * N + 1024 + (gran ? (size_t)16 * N * G : 0);
    char* ws = (ag->xch_ws != nullptr && (size_t)ag->xch_ws_bytes >= need) ? reinterpret_cast<char*>(ag->xch_ws)
               : ((size_t)T * N * sizeof(float) >= need ? reinterpret_cast<char*>(buf->reward_advantages) : nullptr);
    if (!(do_gae & 2) && N <= 128 && N * O <= NORM_CHUNK && O * env->act_dim <= MAX_OBS * MAX_ACT && T >= 1 && ws != nullptr) {
      PersistArgs p;
      p.act = a; p.nm = *nm; p.T = T; p.prof = (do_gae & 4) != 0;
      char* base = ws;
      p.xch_obs = reinterpret_cast<double*>(base); base += (size_t)16 * N * O;
      p.xch_rew = reinterpret_cast<double*>(base); base += (size_t)16 * N;
      p.xch_cost = reinterpret_cast<float*>(base); base += (size_t)8 * N;
      p.xch_done = reinterpret_cast<unsigned*>(base); base += ((size_t)8 * N + 255) / 256 * 256;
      p.counter = reinterpret_cast<unsigned*>(base); base += 512;
      p.xg = reinterpret_cast<unsigned long long*>(base);
      p.g_magic = (unsigned)((1ull << 32) / (unsigned long long)(G / 2));      // index split by the records per env (obs + 2)
      hipError_t e = hipMemsetAsync(p.counter, 0, 512 + (gran ? (size_t)16 * N * G : 0), s);
      if (e != hipSuccess) return (int)e;
      const bool small = a.pl.O <= 32 && (!cn || cn->in_dim <= 32);
      const size_t dyn = persist_dyn_lds(N, O, env->act_dim);
      int coop_err = 0;
      auto go = [&](auto kernel) -> bool {
        if (!persistent_fits(kernel, N, dyn)) return false;
        const hipError_t e_ = launch_coresident(kernel, dim3(N), dim3(256), dyn, s, p);
        if (e_ == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return false; }
        coop_err = (int)e_;
        return true;
      };
      bool launched;
      if (p.prof) {      // (tools: the instantiations with the phase timers)
        if (small && gran) launched = go(rollout_persistent_kernel<2, 2, true, true>);
        else if (small) launched = go(rollout_persistent_kernel<2, 2, false, true>);
        else if (gran) launched = go(rollout_persistent_kernel<8, 10, true, true>);
        else launched = go(rollout_persistent_kernel<8, 10, false, true>);
      } else if (small && gran) launched = go(rollout_persistent_kernel<2, 2, true>);
      else if (small) launched = go(rollout_persistent_kernel<2, 2, false>);
      else if (gran) launched = go(rollout_persistent_kernel<8, 10, true>);
      else launched = go(rollout_persistent_kernel<8, 10, false>);
      int err = coop_err != 0 ? coop_err : (int)hipGetLastError();
      if (!launched) goto per_step;
      if (err || !(do_gae & 1)) return err;
      return icrl_gae_dual_ws(buf->rewards, buf->costs, buf->reward_values, buf->cost_values, buf->dones, ag->last_v_r,
                           ag->last_v_c, ag->last_dones, buf->reward_advantages, buf->cost_advantages, buf->reward_returns,
                           buf->cost_returns, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0, buf->gae_ws,
                          buf->gae_ws_bytes, stream);
    }
  }
per_step:
  for (int t = 0; t < T; ++t) {
    if (a.pl.O <= 32 && (!cn || cn->in_dim <= 32)) hipLaunchKernelGGL((act_step_kernel<2, 2>), dim3(N), dim3(256), 0, s, a, t);
    else hipLaunchKernelGGL((act_step_kernel<8, 10>), dim3(N), dim3(256), 0, s, a, t);
    const size_t row = (size_t)t * N;
    NormStepArgs b{*nm, env->s, ag->raw_rew, cn ? ag->raw_cost : nullptr, ag->dones, N, O, ag->last_obs, nullptr, nullptr,
                   buf->new_observations + row * O, buf->rewards + row, buf->costs + row, ag->last_dones};
    launch_norm_step(b, s);
  }
  int err = (int)hipGetLastError();
  if (err || !(do_gae & 1)) return err;
  return icrl_gae_dual_ws(buf->rewards, buf->costs, buf->reward_values, buf->cost_values, buf->dones, ag->last_v_r,
                       ag->last_v_c, ag->last_dones, buf->reward_advantages, buf->cost_advantages, buf->reward_returns,
                       buf->cost_returns, T, N, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda, 0, buf->gae_ws,
                          buf->gae_ws_bytes, stream);
}

// icrl_rollout_collect for n_runs runs: ONE persistent launch of grid (N, n_runs) + ONE batched dual-GAE launch.  Only the
// one-workgroup-per-env persistent kernel has a batched form (N <= 128 and N x obs <= 4096: BASELINE configs[1]); other shapes
// are refused (the caller then issues the single-run calls).
extern "C" int icrl_rollout_collect_batch(int n_runs, const icrl_rollout_job_t* jobs, const float* action_low, const float* action_high,
                                          double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                                          int do_gae, void* args_ws, long long args_ws_bytes, void* stream) {
  static_assert(sizeof(PersistArgs) <= ICRL_BATCH_ARGS_BYTES, "ICRL_BATCH_ARGS_BYTES");
  if (n_runs < 1 || n_runs > 65535) return fail("icrl_rollout_collect_batch: n_runs = %d (1..65535)", n_runs);
  if (args_ws == nullptr || args_ws_bytes < (long long)n_runs * ICRL_BATCH_ARGS_BYTES)
    return fail("icrl_rollout_collect_batch: args_ws holds %lld B, %d runs need %lld", args_ws_bytes, n_runs, (long long)n_runs * ICRL_BATCH_ARGS_BYTES);
  hipStream_t s = (hipStream_t)stream;
  const icrl_rollout_job_t& j0 = jobs[0];
  const int N = j0.env->n_envs, O = j0.env->obs_dim, T = j0.buf->T;
  const bool has_cn = j0.cn != nullptr;
  // ---- preferred: several envs per workgroup, interleaved (G = N / E workgroups per run: all runs resident together)
  {
    int E = 0, G = 0;
    const int n_stats = O + (has_cn ? 2 : 1);
    static const bool no_multi = getenv("ICRL_BATCH_NO_MULTI") != nullptr;      // tools: the one-workgroup-per-env kernel instead
    // (few small runs: one workgroup per env keeps every env's step at its latency floor and all of them fit the chip at once)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const bool few = (long long)n_runs * N <= cus && N <= 128 && N * O <= NORM_CHUNK;
    if (!no_multi && !few && !j0.pol->discrete && j0.nm->training && N <= NORM_MAX_N && T >= 1 && multi_shape(N, n_stats, &E, &G)) {
      static_assert(sizeof(WideArgs) <= ICRL_BATCH_ARGS_BYTES, "ICRL_BATCH_ARGS_BYTES");
      const size_t GX = 2 * (size_t)O + 4, GS = 4 * (size_t)O + 4;
      const size_t need = 16 * (size_t)N * GX + 16 * GS + 256;
      WideArgs* d_args = (WideArgs*)args_ws;
      bool ok = true;
      for (int r = 0; r < n_runs && ok; ++r) {
        const icrl_rollout_job_t& j = jobs[r];
        if (j.env->n_envs != N || j.env->obs_dim != O || j.env->act_dim != j0.env->act_dim || j.buf->T != T || j.buf->N != N ||
            j.pol->obs_dim != O || j.pol->act_dim != j0.pol->act_dim || j.pol->discrete != j0.pol->discrete || (j.cn != nullptr) != has_cn ||
            (has_cn && (j.cn->in_dim != j0.cn->in_dim || j.cn->n_hidden != j0.cn->n_hidden)) || j.nm->training != j0.nm->training)
          return fail("icrl_rollout_collect_batch: run %d differs from run 0 in a shape (envs / obs / act / T / discrete / constraint net)", r);
        if (!dims_ok(j.pol)) return bad_dims("icrl_rollout_collect_batch", j.pol);
        if (j.buf->obs_dim != O) return fail("icrl_rollout_collect_batch: run %d: buffer obs_dim %d vs env %d", r, j.buf->obs_dim, O);
        if (j.cn != nullptr && !cn_ok(j.cn)) return bad_cn("icrl_rollout_collect_batch", j.cn);
        void* ws = (j.ag->xch_ws != nullptr && (size_t)j.ag->xch_ws_bytes >= need) ? j.ag->xch_ws
                   : ((size_t)T * N * sizeof(float) >= need ? (void*)j.buf->reward_advantages : nullptr);
        if (ws == nullptr) { ok = false; break; }
        WideArgs p;
        ActStepArgs& a = p.act;
        a.env = *j.env; a.buf = *j.buf; a.ag = *j.ag;
        a.pl = make_pol_layout(j.pol->obs_dim, j.pol->act_dim, j.pol->h1, j.pol->h2, j.pol->discrete);
        a.PT = j.pol->params_t; a.noise = j.noise; a.alow = action_low; a.ahigh = action_high;
        a.has_cn = j.cn != nullptr;
        if (j.cn) { a.cn = *j.cn; a.cl = make_cn_layout(j.cn->in_dim, j.cn->n_hidden, j.cn->h1, j.cn->h2); }
        p.nm = *j.nm; p.T = T; p.G = G; p.prof = 0;
        p.xg = reinterpret_cast<unsigned long long*>(ws);
        p.sg = p.xg + 2 * (size_t)N * GX;
        // the workgroups' XCD ids: G words in the 256 spare bytes behind the statistics granules (`need` above)
        p.xcc = G <= 32 ? p.sg + 2 * GS : nullptr;
        hipError_t e = hipMemsetAsync(p.xg, 0, 16 * (size_t)N * GX + 16 * GS + (p.xcc ? 256 : 0), s);
        if (e != hipSuccess) return (int)e;
        const int pe = put_args(p, d_args + r, s);
        if (pe) return pe;
      }
      if (ok) {
        const bool small = O <= 32 && (!has_cn || j0.cn->in_dim <= 32);
        const int err = launch_multi(small, !has_cn || j0.cn->in_dim <= 128, E, nullptr, d_args, n_runs, G, multi_dyn_lds(N, O, j0.env->act_dim, (n_stats + G - 1) / G), s);
        if (err >= 0) {
          if (err || !(do_gae & 1)) return err;
          if (args_ws_bytes < 2ll * n_runs * ICRL_BATCH_ARGS_BYTES)
            return fail("icrl_rollout_collect_batch: args_ws needs 2 x n_runs x ICRL_BATCH_ARGS_BYTES = %lld B when the GAE launch is included", 2ll * n_runs * ICRL_BATCH_ARGS_BYTES);
          return icrl_gae_dual_batch_impl(n_runs, jobs, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda,
                                          (char*)args_ws + (size_t)n_runs * ICRL_BATCH_ARGS_BYTES, stream);
        }
      }
    }
  }
  if (!(N <= 128 && N * O <= NORM_CHUNK && O * j0.env->act_dim <= MAX_OBS * MAX_ACT && T >= 1))
    return fail("icrl_rollout_collect_batch: %d envs x obs %d: no batched form for this shape (multi-env kernel: continuous actions, training statistics, >= 4 envs per statistics-owner group; one workgroup per env: <= 128 envs, envs x obs <= %d)", N, O, NORM_CHUNK);
  const int G = 2 * O + 4;
  const bool gran = (size_t)N * G <= (size_t)256 * GRAN_MAX;
  const size_t need = (size_t)16 * N * O + (size_t)16 * N + (size_t)8 * N + (size_t)8 * N + 1024 + (gran ? (size_t)16 * N * G : 0);
  PersistArgs* d_args = (PersistArgs*)args_ws;
  for (int r = 0; r < n_runs; ++r) {
    const icrl_rollout_job_t& j = jobs[r];
    if (j.env->n_envs != N || j.env->obs_dim != O || j.env->act_dim != j0.env->act_dim || j.buf->T != T || j.buf->N != N ||
        j.pol->obs_dim != O || j.pol->act_dim != j0.pol->act_dim || j.pol->discrete != j0.pol->discrete || (j.cn != nullptr) != has_cn ||
        (has_cn && (j.cn->in_dim != j0.cn->in_dim || j.cn->n_hidden != j0.cn->n_hidden)))
      return fail("icrl_rollout_collect_batch: run %d differs from run 0 in a shape (envs / obs / act / T / discrete / constraint net)", r);
    if (!dims_ok(j.pol)) return bad_dims("icrl_rollout_collect_batch", j.pol);
    if (j.buf->obs_dim != O) return fail("icrl_rollout_collect_batch: run %d: buffer obs_dim %d vs env %d", r, j.buf->obs_dim, O);
    if (j.cn != nullptr && !cn_ok(j.cn)) return bad_cn("icrl_rollout_collect_batch", j.cn);
    char* ws = (j.ag->xch_ws != nullptr && (size_t)j.ag->xch_ws_bytes >= need) ? reinterpret_cast<char*>(j.ag->xch_ws)
               : ((size_t)T * N * sizeof(float) >= need ? reinterpret_cast<char*>(j.buf->reward_advantages) : nullptr);
    if (ws == nullptr) return fail("icrl_rollout_collect_batch: run %d: exchange workspace of %zu B needed (icrl_agent_t.xch_ws, ICRL_ROLLOUT_WS_BYTES)", r, need);
    PersistArgs p;
    ActStepArgs& a = p.act;
    a.env = *j.env; a.buf = *j.buf; a.ag = *j.ag;
    a.pl = make_pol_layout(j.pol->obs_dim, j.pol->act_dim, j.pol->h1, j.pol->h2, j.pol->discrete);
    a.PT = j.pol->params_t; a.noise = j.noise; a.alow = action_low; a.ahigh = action_high;
    a.has_cn = j.cn != nullptr;
    if (j.cn) { a.cn = *j.cn; a.cl = make_cn_layout(j.cn->in_dim, j.cn->n_hidden, j.cn->h1, j.cn->h2); }
    p.nm = *j.nm; p.T = T; p.prof = 0;
    char* base = ws;
    p.xch_obs = reinterpret_cast<double*>(base); base += (size_t)16 * N * O;
    p.xch_rew = reinterpret_cast<double*>(base); base += (size_t)16 * N;
    p.xch_cost = reinterpret_cast<float*>(base); base += (size_t)8 * N;
    p.xch_done = reinterpret_cast<unsigned*>(base); base += ((size_t)8 * N + 255) / 256 * 256;
    p.counter = reinterpret_cast<unsigned*>(base); base += 512;
    p.xg = reinterpret_cast<unsigned long long*>(base);
    p.g_magic = (unsigned)((1ull << 32) / (unsigned long long)(G / 2));      // records per env
    hipError_t e = hipMemsetAsync(p.counter, 0, 512 + (gran ? (size_t)16 * N * G : 0), s);
    if (e != hipSuccess) return (int)e;
    const int pe = put_args(p, d_args + r, s);
    if (pe) return pe;
  }
  const bool small = O <= 32 && (!has_cn || j0.cn->in_dim <= 32);
  const size_t dyn = persist_dyn_lds(N, O, j0.env->act_dim);
  auto go = [&](auto kernel) -> int {
    if (!persistent_fits(kernel, N, dyn)) return fail("icrl_rollout_collect_batch: the %d workgroups of one run do not fit the device", N);
    hipLaunchKernelGGL(kernel, dim3(N, n_runs), dim3(256), dyn, s, d_args);
    return (int)hipGetLastError();
  };
  // registers: the narrow (HC-width) kernel leaves room for several workgroups per CU, so that several runs of the grid are resident
  // together; ICRL_ROLLOUT_MINW (tools) picks the allocation
  // (all workgroups of the grid fit one per CU — up to 4 runs of 64 envs: the full register file per workgroup, 17 ms per 2048-step
  // rollout against 29.6 with the two-per-CU allocation, measured at S = 1 and 4)
  static const int minw_env = getenv("ICRL_ROLLOUT_MINW") ? atoi(getenv("ICRL_ROLLOUT_MINW")) : 0;
  int dev_ = 0, cus_ = 256;
  if (hipGetDevice(&dev_) != hipSuccess || hipDeviceGetAttribute(&cus_, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess) cus_ = 256;
  const int minw = minw_env > 0 ? minw_env : ((long long)n_runs * N <= cus_ ? 1 : 2);
  int err;
  if (small && gran) err = minw >= 3 ? go(rollout_persistent_batch_kernel<2, 2, true, 3>) : (minw == 2 ? go(rollout_persistent_batch_kernel<2, 2, true, 2>) : go(rollout_persistent_batch_kernel<2, 2, true, 1>));
  else if (small) err = go(rollout_persistent_batch_kernel<2, 2, false, 2>);
  else if (gran) err = go(rollout_persistent_batch_kernel<8, 10, true, 1>);
  else err = go(rollout_persistent_batch_kernel<8, 10, false, 1>);
  if (err || !(do_gae & 1)) return err;
  // dual GAE of every run in one launch; its argument blocks go behind the rollout's in args_ws (both launches are in flight together)
  if (args_ws_bytes < 2ll * n_runs * ICRL_BATCH_ARGS_BYTES)
    return fail("icrl_rollout_collect_batch: args_ws needs 2 x n_runs x ICRL_BATCH_ARGS_BYTES = %lld B when the GAE launch is included", 2ll * n_runs * ICRL_BATCH_ARGS_BYTES);
  return icrl_gae_dual_batch_impl(n_runs, jobs, reward_gamma, reward_gae_lambda, cost_gamma, cost_gae_lambda,
                                  (char*)args_ws + (size_t)n_runs * ICRL_BATCH_ARGS_BYTES, stream);
}

extern "C" int icrl_rollout_collect(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol,
                                    const icrl_costnet_t* cn, const icrl_buffer_t* buf, const icrl_agent_t* ag,
                                    const float* noise, const float* action_low, const float* action_high,
                                    double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                                    void* stream) {
  return icrl_rollout_collect_ex(env, nm, pol, cn, buf, ag, noise, action_low, action_high, reward_gamma, reward_gae_lambda,
                                 cost_gamma, cost_gae_lambda, 1, stream);
}
