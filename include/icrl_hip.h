/* libicrl_hip.so — C ABI of the MI355X-native ICRL rollout+update hot path.
 *
 * The reference (shehryar-malik/icrl) is pure Python: its "FFI" for this path is the set of numpy / torch
 * calls made by the classes listed next to each entry point below (paths relative to /root/reference).
 * Each function here replaces the arithmetic of one such call site with a hand-written gfx950 kernel.
 *
 * Conventions (all entry points):
 *   - every pointer is DEVICE memory owned by the caller (torch-allocated HBM in the Python host);
 *     the library never allocates or frees user-visible memory;
 *   - `stream` is a hipStream_t (passed as void* so that the header needs no HIP include); work is
 *     enqueued asynchronously on it, there is no hidden synchronisation;
 *   - the return value is a hipError_t as int (0 = hipSuccess); nothing throws or aborts;
 *   - one host thread per GPU / process; re-entrant across devices, not thread-safe on shared buffers;
 *   - [T,N,...] arrays are time-major, contiguous, float32 unless stated (RolloutBufferWithCost layout,
 *     stable_baselines3/common/buffers.py:468-491).
 */
#ifndef ICRL_HIP_H
#define ICRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library / build identification: returns ABI version (major*100+minor). */
int icrl_abi_version(void);

/* Dual reward+cost GAE over a [T,N] rollout in ONE launch.
 * Replaces RolloutBufferWithCost.compute_returns_and_advantage / _compute_returns_and_advantage
 *   (stable_baselines3/common/buffers.py:493-552), including its dtype behaviour: float32 delta for t<T-1,
 *   float64 running advantage, float32 rounding on store, returns = adv + values in float32.
 * dones[t,n] is the done flag ENTERING step t (what the buffer stores); last_dones[n] is the final step's done flag (0/1 bytes).
 * last_v_r / last_v_c: [N] bootstrap values.  Outputs adv_*, ret_*: [T,N].
 * Algorithmic traffic: 36 B per transition (5 loads + 4 stores of 4 B). */
int icrl_gae_dual(const float* rewards, const float* costs, const float* reward_values, const float* cost_values,
                  const float* dones, const float* last_v_r, const float* last_v_c, const uint8_t* last_dones,
                  float* adv_r, float* adv_c, float* ret_r, float* ret_c,
                  int T, int N, double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                  void* stream);

/* Same kernel with the launch shape forced (for the roofline sweep in bench.py / tests):
 * waves_per_tile in {1,4,16} time-chunks per 64-env column tile; 0 = library heuristic. */
int icrl_gae_dual_ex(const float* rewards, const float* costs, const float* reward_values, const float* cost_values,
                     const float* dones, const float* last_v_r, const float* last_v_c, const uint8_t* last_dones,
                     float* adv_r, float* adv_c, float* ret_r, float* ret_c,
                     int T, int N, double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                     int waves_per_tile, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICRL_HIP_H */
