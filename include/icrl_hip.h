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
 *   - structs are plain C, passed by pointer to HOST memory (they hold device pointers + scalars) and are
 *     only read during the call.
 *
 * Co-residency precondition of the PERSISTENT launches (icrl_rollout_collect[_ex|_batch], icrl_ppo_lag_train[_batch]): the
 * workgroups of ONE run wait for each other inside the launch (granule exchange), so they must all be resident at the same
 * time: n_envs workgroups of 256 threads for the rollout (<= 128 envs; the many-environment kernel sizes its grid to what
 * hipOccupancyMaxActiveBlocksPerMultiprocessor reports), 3 (6 when a minibatch is two chunks at obs_dim > 64) workgroups for
 * the update, each filling a CU's LDS.  The SINGLE-RUN entry points issue these grids with hipLaunchCooperativeKernel (round 4): the
 * runtime refuses a grid it cannot make co-resident (the rollout then falls back to per-step launches), after the library's own
 * check of the occupancy the runtime reports; the *_batch grids are ordinary launches.  Whatever else occupies CUs for long at the same time —
 * another stream of this process, or ANOTHER PROCESS on the same GPU, which no host-side check of this process can see — can
 * keep a workgroup of a run from being scheduled; its peers then spin (bounded: ~2 s per exchange, s_sleep between polls), the
 * launch ends with icrl_agent_t.status bit 0 / stats[11] set and the host raises — buffers and statistics of that call are
 * invalid, nothing hangs.  Runs of one *_batch grid do not wait for each other: a run whose workgroups do not fit yet starts
 * when earlier runs of the grid have finished (dispatch is in grid order).
 * XCD placement (speed only): the update launches and the batched multi-env rollout are 1-D grids in which the workgroups of run r sit
 * at 8 M (r / 8) + r % 8 + 8 j, j < M (M = 3 | 6 | workgroups of the rollout; surplus workgroups of the grid leave at once) — workgroups are
 * dealt round-robin over the 8 XCDs, so a run then sits on ONE XCD, and granules stored with workgroup scope stay in that XCD's L2 for
 * the other workgroups' polls.  Nothing relies on it: every workgroup publishes its HW_REG_XCC_ID once per launch and the workgroup-scope
 * stores are used only where all workgroups of a run reported the same XCD; batched grids too large for every XCD to hold its share at
 * once (32 CUs) keep the run-major layout (3 | 6 | G, n_runs) and agent-scope stores.
 */
#ifndef ICRL_HIP_H
#define ICRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Data descriptors
 * ------------------------------------------------------------------------------------------------------------------ */

/* Synthetic HCWithPos- / AntWall-shaped vectorised environment (oracle/synth_env.py is the spec).
 * Stands in for SubprocVecEnv + gym envs (stable_baselines3/common/vec_env/subproc_vec_env.py:14-50,104-113;
 * custom_envs/envs/half_cheetah.py:138-195, ant.py:40-108): float64 state, time-limit dones, auto-reset. */
typedef struct {
  int32_t n_envs, obs_dim, act_dim, max_steps;
  int32_t reward_form;    /* 0: |dx|/dt - 0.1|a|^2 (HC)   1: |xy| + 1 - 0.5|a|^2 (Ant)
                           * 2: LapGridWorld, 3: ConstrainedLapGridWorld (custom_envs/envs/lap_grid_world.py:62-240;
                           *    obs_dim 1, act_dim 1 = the Discrete(2) action index as a float; B / key unused) */
  int32_t wall_terminate; /* "Test" variants: done & reward 0 when obs[0] <= -3 */
  int32_t broken;         /* AntWallBroken: action[4:] = 0 */
  int32_t _pad;
  const double* B;        /* [obs_dim, act_dim] dynamics matrix */
  double* s;              /* [N, obs_dim] raw state == last raw observation (== VecCostWrapper.previous_obs) */
  int32_t* t_ep;          /* [N] step inside the episode */
  uint32_t* step_count;   /* [N] counter of the per-env random stream */
  const uint32_t* key;    /* [N] stream key = seed + global env index */
} icrl_env_t;

/* VecNormalizeWithCost state (stable_baselines3/common/vec_env/vec_normalize.py:9-181,184-278;
 * RunningMeanStd: common/running_mean_std.py:6-39).  All float64 as in the reference. */
typedef struct {
  int32_t training, norm_obs, norm_reward, norm_cost;
  double clip_obs, clip_reward, clip_cost, reward_gamma, cost_gamma, epsilon;
  double* obs_mean;   /* [obs_dim] */
  double* obs_var;    /* [obs_dim] */
  double* obs_count;  /* [1] */
  double* ret_stats;  /* [3] mean, var, count of ret_rms */
  double* cost_stats; /* [3] mean, var, count of cost_rms */
  double* ret;        /* [N] discounted reward return */
  double* cost_ret;   /* [N] discounted cost return */
} icrl_norm_t;

/* ActorTwoCriticsPolicy parameters (stable_baselines3/common/policies.py:598-779): three tanh MLPs obs->h1->h2
 * (pi, vf, cvf), heads h2->act / 1 / 1, state-independent log_std.  `params` is ONE flat float32 buffer in the
 * reference's state_dict order: log_std[A] (absent when discrete), then for pi, vf, cvf: W1[h1,obs] b1[h1] W2[h2,h1]
 * b2[h2], then action_net W[A,h2] b[A], value_net W[1,h2] b[1], cost_value_net W[1,h2] b[1].
 *
 * Any other MlpExtractor architecture (common/torch_layers.py:129-254: `-sl` shared trunk, `-pl / -rvl / -cvl` of any depth) is
 * described by `arch`, HOST memory: { n_shared, widths..., n_pi, widths..., n_vf, widths..., n_cvf, widths... } with 0..4 layers per
 * group and 1..256 units per layer (a branch without layers feeds its head from the trunk, or from the observation).  h1 / h2 are then
 * ignored and `params` holds the natural, unpadded layout in state_dict order: log_std, the trunk's layers (W[out,in] b[out] each),
 * policy_net's, value_net's, cost_value_net's, then the three heads.  Such a policy is served by icrl_policy_forward,
 * icrl_policy_evaluate, icrl_ppo_lag_train, icrl_rollout_collect(_ex) and icrl_sample_episodes (generic-shape path: one persistent launch
 * per update / rollout / sampling call like the fast path, slower per step — DESIGN.md section 8); the *_batch entry points refuse it. */
typedef struct {
  int32_t obs_dim, act_dim, h1, h2;
  int32_t discrete; /* 1: Categorical over act_dim logits (LGW), 0: DiagGaussian */
  int32_t n_params;
  float* params;     /* [n_params] */
  float* params_t;   /* [n_params] transposed copy for the forward kernels (written by icrl_policy_prepare; icrl_ppo_lag_train keeps it current) */
  const int32_t* arch; /* NULL: two hidden layers per branch (h1, h2), no shared trunk */
} icrl_policy_t;

/* ConstraintNet zeta_theta (icrl/constraint_net.py:14-130,258-299): ReLU MLP + sigmoid over
 * concat(clip(obs), clip(acs))[select_dim]; cost = 1 - zeta.  n_hidden hidden layers of h1 .. h4 units (`-cl`, torch_layers.py:93-126):
 * 1 or 2 layers of up to 64 units inside the fused rollouts; 0..4 layers through icrl_cost_mlp_forward / icrl_disc_reward /
 * icrl_cn_train* (64 rows per workgroup) as long as the 64-row activation images fit the 160 KB LDS — e.g. 2 x 128, 3 x 96, 4 x 64 units
 * (refused with the byte count otherwise).  params flat in state_dict order: W0[h1,in] b0[h1] (W1[h2,h1] b1[h2] ...) Wo[1,h] bo[1]. */
typedef struct {
  int32_t obs_dim, acs_dim, in_dim, n_hidden, h1, h2, h3, h4;
  int32_t is_discrete;  /* one-hot the action first */
  int32_t n_params;
  double clip_obs;           /* < 0: no clipping (what ConstraintNet.load builds, constraint_net.py:394-399) */
  const int32_t* select_dim; /* [in_dim] indices into concat(obs, acs) */
  const float* action_low;   /* [acs_dim] or NULL: no action clipping */
  const float* action_high;  /* [acs_dim] or NULL */
  const double* obs_mean;    /* [obs_dim] or NULL (--cn_normalize) */
  const double* obs_var;     /* [obs_dim] or NULL */
  double eps;
  float* params;             /* [n_params] */
  float* params_t;           /* [n_params] transposed copy (written by icrl_costnet_prepare) */
} icrl_costnet_t;

/* RolloutBufferWithCost arrays (stable_baselines3/common/buffers.py:443-491), time-major [T,N,...] float32. */
typedef struct {
  int32_t T, N, obs_dim, act_store; /* act_store = act_dim (Box) or 1 (Discrete) */
  float *observations, *new_observations, *orig_observations, *new_orig_observations; /* [T,N,obs] */
  float* actions;                                                                      /* [T,N,act_store] */
  float *dones, *log_probs, *rewards, *reward_values, *costs, *orig_costs, *cost_values;
  float *reward_advantages, *reward_returns, *cost_advantages, *cost_returns; /* [T,N] */
  void* gae_ws;             /* optional workspace of the GAE launch (see icrl_gae_dual_ws): device memory, zeroed once when */
  long long gae_ws_bytes;   /* allocated, afterwards written by GAE launches only.  NULL: one workgroup per column tile */
} icrl_buffer_t;
#define ICRL_GAE_WS_BYTES (256 * (256 * 8 + 4) + 64)   /* 256 maps + flags, + the status word */

/* What OnPolicyWithCostAlgorithm carries between steps (common/on_policy_algorithm.py:367-416, base_class.py:346-353)
 * plus per-step scratch. */
typedef struct {
  double* last_obs;      /* [N,obs] normalised observation fed to the policy (== _last_obs) */
  uint8_t* last_dones;   /* [N] */
  double* raw_rew;       /* [N] scratch: un-normalised reward of the step (== get_original_reward) */
  float* raw_cost;       /* [N] scratch: zeta-cost of the step (== get_original_cost) */
  uint8_t* dones;        /* [N] scratch: done flags of the step */
  float* last_v_r;       /* [N] values of the last forward (GAE bootstrap, on_policy_algorithm.py:417) */
  float* last_v_c;       /* [N] */
  float* act_clipped;    /* [N,act] scratch: action handed to the env */
  int* status;           /* [1] or NULL: the persistent rollout ORs bit 0 into it when an inter-workgroup exchange timed out
                          * (a workgroup was not resident); the buffer contents are then invalid and the caller must raise */
  void* xch_ws;          /* optional exchange workspace of the persistent rollout (device memory, any contents) ... */
  long long xch_ws_bytes;/* ... and its size; >= ICRL_ROLLOUT_WS_BYTES(N, obs) suffices for both persistent kernels.  NULL / too small: the not yet
                          * computed reward_advantages plane of the buffer is used when T is large enough, else per-step launches */
} icrl_agent_t;
#define ICRL_ROLLOUT_WS_BYTES(n_envs, obs_dim) (48 * (size_t)(n_envs) * (size_t)(obs_dim) + 96 * (size_t)(n_envs) + 64 * (size_t)(obs_dim) + 2048)

/* PPO-Lagrangian update hyper-parameters (stable_baselines3/ppo_lag/ppo_lag.py:67-103,177-196; Adam eps 1e-5 from
 * common/policies.py:357-361). */
typedef struct {
  int32_t batch_size, n_epochs, use_target_kl, _pad;
  float clip_range, ent_coef, reward_vf_coef, cost_vf_coef, max_grad_norm, target_kl;
  float clip_range_reward_vf, clip_range_cost_vf; /* < 0: no value clipping (the default) */
  float lr, adam_beta1, adam_beta2, adam_eps;
} icrl_ppo_hyper_t;

/* ConstraintNet.train hyper-parameters (icrl/constraint_net.py:15-42,137-229). */
typedef struct {
  int32_t iterations, importance_sampling, per_step, gail;
  float reg_coeff, eps, target_kl_old_new, target_kl_new_old; /* target < 0 (== -1): that KL test is disabled */
  float lr, adam_beta1, adam_beta2, adam_eps;
} icrl_cn_hyper_t;

#define ICRL_PPO_SPLIT_BYTES (2560 * 1024) /* exchange space of the multi-workgroup updates: partial gradients (2 step parities x 3 networks x up to 4 parts), updated parameters */
/* granule slots (512 B) + the schedule tables: 16 B per optimiser step and 8 B per 64-row chunk (up to four per step) */
#define ICRL_PPO_PLAN_BYTES(n_steps) (768 + 48 * (size_t)(n_steps))
#define ICRL_PPO_SYNC_BYTES(n_epochs, n_minibatches, n_rows) \
  (ICRL_PPO_PLAN_BYTES((size_t)(n_epochs) * (size_t)(n_minibatches)) + 4 * (size_t)(n_epochs) * (size_t)(n_rows) + 256 + ICRL_PPO_SPLIT_BYTES)
/* generic-shape update (hidden widths above 64, a policy with `arch`, or batch_size above 256): its scratch lies BEHIND the regular
 * workspace, sync_ws then holds ICRL_PPO_SYNC_BYTES(...) + ICRL_PPO_GENERIC_BYTES(...) bytes.  row_floats = the outputs of every layer
 * of one row = icrl_ppo_generic_row_floats(pol) (6 h + act_dim + 2 for the padded two-layer layout of common width h) */
#define ICRL_PPO_GENERIC_BYTES(batch_size, row_floats, n_params) \
  (4 * (64 + (size_t)(batch_size) * (24 + 1 + 16 + 2 * (size_t)(row_floats)) + (size_t)(n_params) + ((size_t)(n_params) + 255) / 256 + 1088 + \
        ICRL_PPO_GENERIC_PERSIST_FLOATS(batch_size, n_params)))
/* the one-launch form of the generic-shape update (up to 32 row tiles of 16 rows, up to 131 072 parameters): per-tile partial gradients */
#define ICRL_PPO_GENERIC_PERSIST_FLOATS(batch_size, n_params) \
  ((((batch_size) + 15) / 16 <= 32 && (n_params) <= 131072) ? (size_t)(((batch_size) + 15) / 16 + 1) * (size_t)(n_params) + 1024 : (size_t)0)
#define ICRL_CN_METRICS 24 /* floats per iteration in the metrics array of icrl_cn_train */

/* ------------------------------------------------------------------------------------------------------------------
 * Entry points
 * ------------------------------------------------------------------------------------------------------------------ */

/* ABI version (major*100+minor). */
int icrl_abi_version(void);   /* 106: icrl_explained_variance, icrl_gae_dual_ws_bytes + the status word of the GAE workspace; 105: icrl_buffer_add, icrl_is_weights, icrl_cn_loss_fwd_bwd; 104: the fine-grained update entry points (csrc/fine.hip); 103: icrl_debug_stream_ref; 102: icrl_policy_t.arch; 101: icrl_sample_episodes takes stream_row0 / total_rows; the *_batch entry points */

/* Every entry point returns a hipError_t.  When it is hipErrorInvalidValue because the arguments are outside what the
 * kernels were built for (env count, widths, batch size ...; the reference's Python raises ValueError / AssertionError with a
 * message at the same places, e.g. stable_baselines3/common/buffers.py:352, on_policy_algorithm.py:141), the reason is kept
 * as text for the calling host thread until the next refusal; icrl_clear_error() empties it. */
const char* icrl_last_error(void);
void icrl_clear_error(void);

/* Dual reward+cost GAE over a [T,N] rollout in ONE launch.
 * Replaces RolloutBufferWithCost.compute_returns_and_advantage / _compute_returns_and_advantage
 *   (stable_baselines3/common/buffers.py:493-552), including its dtype behaviour: float32 delta for t<T-1,
 *   float64 running advantage, float32 rounding on store, returns = adv + values in float32.
 * dones[t,n] is the done flag ENTERING step t (what the buffer stores); last_dones[n] the final step's flag (0/1 bytes).
 * Algorithmic traffic: 36 B per transition (5 loads + 4 stores of 4 B). */
int icrl_gae_dual(const float* rewards, const float* costs, const float* reward_values, const float* cost_values,
                  const float* dones, const float* last_v_r, const float* last_v_c, const uint8_t* last_dones,
                  float* adv_r, float* adv_c, float* ret_r, float* ret_c,
                  int T, int N, double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                  void* stream);

/* Same kernel with the launch shape forced (roofline sweep): waves_per_tile in {1,4,16}; 0 = library heuristic;
 * 101 / 105 / 106 = one-wave-per-tile shapes with 1 / 4 / 4 tiles per workgroup and 8 / 8 / 16 rows in flight;
 * 107..112 = four columns per lane (16-byte accesses; N % 4 == 0): 2 / 4 / 8 rows in flight with 4 waves (107-109) or 1 wave
 * (110-112) per workgroup.  All 1xx shapes are bit-exact replicas of the sequential scan. */
int icrl_gae_dual_ex(const float* rewards, const float* costs, const float* reward_values, const float* cost_values,
                     const float* dones, const float* last_v_r, const float* last_v_c, const uint8_t* last_dones,
                     float* adv_r, float* adv_c, float* ret_r, float* ret_c,
                     int T, int N, double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                     int waves_per_tile, void* stream);

/* The same scan with a caller-owned workspace (zeroed once at allocation and written by these launches only).  Up to 65 472 envs and
 * T <= 2048 (round 6): the REGISTER-RESIDENT split scan — ceil(T / 128) workgroups per 64-env column tile, each wave keeps its 16 rows in
 * registers between the two passes of a two-level scan over affine maps, so every byte is read once (the BASELINE-size launch, 64 envs x
 * 2048 rows, walks 16 rows per wave instead of 128; 8 192 .. 65 472 envs get 2 048 .. 16 368 workgroups instead of one wave per tile).
 * It needs icrl_gae_dual_ws_bytes(T, N) bytes; ICRL_GAE_WS_BYTES suffices for up to 1024 envs at T <= 2048.  With less (or T > 2048) and up
 * to 128 column tiles: the older two-pass split over up to 16 workgroups per tile.
 * The LAST 4 bytes of the workspace (of its size rounded down to 8) are the launch's STATUS word: every wait for another workgroup's map is
 * bounded (~seconds); a map that never arrives ends the wait with status = 1 and invalid outputs — the caller checks and clears it.
 * waves_per_tile as above, or 200 + C to force the two-pass split with C workgroups per tile, 500 the register-resident split; 300 + C and
 * 501 are those two with one map withheld and a short limit (fault injection for the tests).  ws == NULL: identical to icrl_gae_dual_ex. */
size_t icrl_gae_dual_ws_bytes(int T, int N);
int icrl_gae_dual_ws(const float* rewards, const float* costs, const float* reward_values, const float* cost_values,
                     const float* dones, const float* last_v_r, const float* last_v_c, const uint8_t* last_dones,
                     float* adv_r, float* adv_c, float* ret_r, float* ret_c,
                     int T, int N, double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                     int waves_per_tile, void* ws, long long ws_bytes, void* stream);

/* Write the transposed ([in][out]) weight copies the rollout-time kernels read coalesced.  Call after every change of
 * `params` (policy: after train(); cost net: after ConstraintNet.train / load). */
int icrl_policy_prepare(const icrl_policy_t* pol, void* stream);
int icrl_costnet_prepare(const icrl_costnet_t* cn, void* stream);

/* ActorTwoCriticsPolicy.forward (policies.py:716-731) for N observations: sample (mean + noise*std, or inverse-CDF on
 * `noise` uniforms when discrete; deterministic: mean / argmax), log-prob, both values.
 * obs: [N,obs] float64 (cast to float32 like preprocess_obs, preprocessing.py:61).  noise: [N,act] standard normals
 * ([N] uniforms when discrete) or NULL when deterministic.  actions: [N,act_store] unclipped; act_clipped: [N,act]
 * clipped to [low,high] (on_policy_algorithm.py:381-382; NULL low/high: no clipping).  Any output may be NULL.
 * Policies stored with h1 = h2 > 64 (a multiple of 64 up to 256; obs up to 1024) or described by `arch` run the generic-shape kernel
 * here and in icrl_policy_evaluate (it reads the per-layer transposes icrl_policy_prepare leaves in params_t); icrl_rollout_collect(_ex)
 * then runs the whole rollout as ONE persistent launch around that forward (up to 128 envs with a constraint net of at most two layers
 * of 64 units; otherwise the reference's per-step loop as four launches per step: policy forward, constraint-net cost, env step,
 * normaliser) and icrl_sample_episodes runs its episode loop with that forward;
 * the *_batch entry points refuse such shapes with a message. */
int icrl_policy_forward(const icrl_policy_t* pol, const double* obs, const float* noise, int N, int deterministic,
                        const float* action_low, const float* action_high,
                        float* actions, float* act_clipped, float* v_r, float* v_c, float* log_prob, void* stream);

/* same, with options in `do_gae`: bit 0 = run the final dual-GAE launch (0: the caller launches icrl_gae_dual itself, e.g.
 * bracketed by events); bit 1 = force one launch pair per env step; bit 2 = diagnostic phase timers.  By default, when N <= 128 and N * obs_dim <= 4096, ALL
 * T steps run in ONE persistent launch (one workgroup per env, one device-wide barrier per step, normaliser statistics
 * replicated bit-identically in every workgroup); the not yet computed reward_advantages plane of the buffer serves as its
 * exchange scratch. */
int icrl_rollout_collect_ex(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol, const icrl_costnet_t* cn,
                            const icrl_buffer_t* buf, const icrl_agent_t* ag, const float* noise,
                            const float* action_low, const float* action_high,
                            double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                            int do_gae, void* stream);

/* ActorTwoCriticsPolicy.evaluate_actions (policies.py:752-767): values, log-prob of the GIVEN actions and the entropy of
 * the action distribution (used by compute_kl, icrl/utils.py:421-437).  actions [N,act_store] float32. */
int icrl_policy_evaluate(const icrl_policy_t* pol, const double* obs, const float* actions, int N, float* v_r, float* v_c,
                         float* log_prob, float* entropy, void* stream);

/* sample_from_agent (icrl/utils.py:323-357) / evaluate_policy (stable_baselines3/common/evaluation.py:10-67) on single-env
 * streams, one persistent workgroup per stream: env->n_envs independent streams each run `episodes_per_stream` episodes
 * back to back (auto-reset between them, like a 1-env VecEnv) with FROZEN normaliser statistics (nm->training must be 0).
 * Row k of a stream records the observation AFTER step k (raw in orig_obs, normalised in obs) next to the clipped action
 * of step k — the (s_{t+1}, a_t) pairing of the reference — at rows [stream*rows_per_stream + k].  noise: standard normals
 * [n_streams*rows_per_stream, act] (NULL or deterministic != 0: mode of the distribution).  do_reset: draw the initial
 * state first (VecEnv.reset).  Per-episode un-normalised reward sums and lengths go to ep_rewards / ep_lengths.
 * stream_row0 ([n_streams] int32 on the device, or NULL): first row of every stream in the noise and output arrays instead of
 * stream * rows_per_stream, total_rows = rows of those arrays.  This is how episodes of a sequential 1-env loop that may end
 * early (the "Test" envs) run as parallel streams: the host guesses every episode's position in the loop (its start row = the
 * steps taken before it, which is also its position in the env's random stream, env->step_count), runs all of them, and
 * repeats with the positions the measured lengths imply until they agree; rows are then exactly the sequential loop's. */
int icrl_sample_episodes(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol, const float* noise,
                         const float* action_low, const float* action_high, int episodes_per_stream, int rows_per_stream,
                         int deterministic, int do_reset, const int32_t* stream_row0, int total_rows, double* orig_obs,
                         double* obs, float* actions, double* ep_rewards, int32_t* ep_lengths, void* stream);

/* ConstraintNet.cost_function (icrl/constraint_net.py:121-130): cost[n] = 1 - zeta(prepare(obs[n], acs[n])).
 * obs [N,obs] float64, acs [N,acs] float32 (class index in acs[n,0] when discrete). */
int icrl_cost_mlp_forward(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* cost, void* stream);

/* GailDiscriminator.reward_function (icrl/gail_utils.py:147-157): the same network read the other way round — the
 * discriminator output D = sigmoid(MLP(prepare(obs, acs))) itself (apply_log = 0) or log(D + eps) (apply_log = 1, the
 * GAIL-constraint baseline's reward, eps = cn->eps).  out: [N] float32. */
int icrl_disc_reward(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* out, int apply_log,
                     void* stream);

/* SynthVecEnv.reset / .step: the batched stand-in for SubprocVecEnv.step_async/step_wait.  step: actions [N,act] float32
 * (already clipped), writes raw reward [N] float64 and done [N] bytes, advances env->s in place (auto-reset). */
int icrl_synth_env_reset(const icrl_env_t* env, void* stream);
int icrl_synth_env_step(const icrl_env_t* env, const float* actions, double* raw_rew, uint8_t* dones, void* stream);

/* VecNormalizeWithCost.reset (vec_normalize.py:148-157,270-278): zero ret / cost_ret, feed a zero batch into ret_rms /
 * cost_rms when training, write normalize_obs(raw_obs) to obs_out [N,obs] float64. */
int icrl_vecnorm_reset(const icrl_norm_t* nm, const double* raw_obs, int N, int obs_dim, double* obs_out, void* stream);

/* VecNormalizeWithCost.step_wait after the env + cost wrapper (vec_normalize.py:81-100,220-243): running-moment merge
 * (obs, discounted reward return, discounted cost return), normalise + clip, zero the returns of finished envs.
 * raw_cost may be NULL (env stack without cost wrapper).  Outputs float64 [N,obs] / [N] / [N] (cost_out may be NULL). */
int icrl_vecnorm_step(const icrl_norm_t* nm, const double* raw_obs, const double* raw_rew, const float* raw_cost,
                      const uint8_t* dones, int N, int obs_dim, double* obs_out, double* rew_out, double* cost_out,
                      void* stream);

/* OnPolicyWithCostAlgorithm.collect_rollouts (common/on_policy_algorithm.py:340-421) fused on the device: T steps of
 * {policy forward -> clip -> env step -> cost_function(previous raw obs, action) -> VecNormalizeWithCost -> buffer.add},
 * then the dual GAE.  noise: [T,N,act] standard normals.  ONE persistent launch for all T steps where the shape allows
 * (see icrl_rollout_collect_ex: <= 128 envs one workgroup per env with replicated statistics, up to 1024 envs 4 envs per
 * workgroup with the float64 statistics partitioned by observation column), otherwise 2 launches per step; + 1 launch for
 * GAE; no host sync. */
int icrl_rollout_collect(const icrl_env_t* env, const icrl_norm_t* nm, const icrl_policy_t* pol, const icrl_costnet_t* cn,
                         const icrl_buffer_t* buf, const icrl_agent_t* ag, const float* noise,
                         const float* action_low, const float* action_high,
                         double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                         void* stream);

/* PPOLagrangian.train (stable_baselines3/ppo_lag/ppo_lag.py:196-299) as ONE persistent launch: for every epoch, for every
 * minibatch of the permutation: gather (buffers.py:594-627, env-major flat index i -> env = i / T, t = i % T),
 * evaluate_actions (policies.py:752-767), advantage normalisation / centring, clipped surrogate + nu * cost term, two value
 * MSEs, entropy, backward, global-norm clip (max_grad_norm), Adam, approx-KL early stop after a full epoch.
 * Three workgroups (policy | reward critic | cost critic — the three MLPs are independent) keep their weights in LDS and
 * their Adam moments in registers; the only coupling per step is the squared gradient norm (3 floats, 8-byte granules).
 *   exp_avg / exp_avg_sq: [n_params] Adam moments (policy.optimizer state), adam_step: [1] int32 step counter (device);
 *   perms: [n_epochs, T*N] int32 permutations;  nu: [1] current Lagrange multiplier (device);
 *   stats: [32 + n_epochs] float32 (device): 0 early_stop_epoch, 1 optimiser steps done, 2..6 sums over minibatches of
 *          entropy_loss / policy_loss / reward_value_loss / cost_value_loss / clip_fraction, 7 mean approx_kl of the last
 *          epoch, 8..10 last minibatch's policy+entropy / reward-value / cost-value loss terms, 11 status (0 ok),
 *          32+e mean approx_kl of epoch e;
 *   sync_ws: >= ICRL_PPO_SYNC_BYTES(n_epochs, ceil(T*N / batch_size), T*N) bytes of device scratch: the inter-workgroup granules
 *            (zeroed by the call) + the schedule tables the call fills (per optimiser step: minibatch rows, Adam bias
 *            corrections; per 64-row chunk: permutation offset; the permutations mapped to storage offsets).
 *   hp->_pad: bit 0 = write per-phase cycle counts to stats[12..31] (diagnostic); kernel selection (default: wave pairs, two
 *            waves per SIMD; obs > 64: row-owning waves): bit 2 = row-owning waves (one wave per SIMD), bit 1 = the column-split
 *            tiles kernel; the row-owning kernel runs TWO workgroups per network when batch_size > 64 (each computes one 64-row
 *            chunk of a minibatch, partial gradients exchanged as granules) unless bit 3 is set.  obs <= 32 (HCWithPos, LapGridWorld): the
 *            wave-quad kernel — FOUR workgroups per network (12 per run on one XCD; 16 rows of every 64-row chunk each, the four partial
 *            gradients summed as (q0 + q1) + (q2 + q3) by all four; round 6), TWO with bit 5 (32 rows each; also what batches of 17..40
 *            runs get), the wave-pair kernel (one) with bit 4.  obs 65..128 (AntWall), single-run calls: FOUR workgroups per network as well
 *            (wave quads with the row-owning kernel's parameter ownership; minibatches of 65..128 rows take both 64-row chunks in one pass);
 *            bit 2 or bit 5 keep the row-owning kernel there.  Bit 0 selects separate instantiations of the kernels (the timers cost every
 *            launch ~2.5 % as a run-time flag).  Bit 6 (tests): the last workgroup of a four-workgroups-per-network launch leaves at once —
 *            the others' bounded waits must end the launch with stats[11] set.
 * Shapes outside the persistent kernels — a policy stored with hidden width h1 = h2 > 64 (a multiple of 64 up to 256: the reference's
 * -pl / -rvl / -cvl flags take any width, icrl/utils.py:636-655), a policy described by `arch` (shared trunk, other depths) or
 * batch_size > 256 (buffers.py:594-612 slices any size) — run through the generic-shape path (csrc/generic.hip): ONE persistent
 * cooperative launch on one XCD (up to 512-row batches, 131 072 parameters, rows that fit the LDS; otherwise three plain launches per
 * optimiser step), same statistics layout, of hp->_pad only bit 0 (phase timers in stats[12..28]) and bit 6 (tests: one workgroup leaves, the bounded waits must end the launch with stats[11] set); sync_ws must then hold ICRL_PPO_SYNC_BYTES(...) +
 * ICRL_PPO_GENERIC_BYTES(batch_size, icrl_ppo_generic_row_floats(pol), n_params) bytes. */
int icrl_ppo_lag_train(const icrl_policy_t* pol, float* exp_avg, float* exp_avg_sq, int32_t* adam_step,
                       const icrl_buffer_t* buf, const int32_t* perms, const float* nu, const icrl_ppo_hyper_t* hp,
                       float* stats, void* sync_ws, void* stream);
/* the `row_floats` argument of ICRL_PPO_GENERIC_BYTES for this policy (host arithmetic, no launch), or -1 with the reason in
 * icrl_last_error() when the generic-shape path does not serve its architecture */
int icrl_ppo_generic_row_floats(const icrl_policy_t* pol);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched forms: several INDEPENDENT runs (seeds) of identical shape in ONE launch, run = blockIdx.y
 * ------------------------------------------------------------------------------------------------------------------
 * One run of BASELINE configs[1] occupies 3 CUs during its update and 64 during a rollout; the reference runs seeds as separate
 * processes (README.md:14-21, one `python run_me.py` per seed).  Here n_runs runs share one grid: every *_batch entry point takes
 * an array of n_runs job descriptors in HOST memory (same fields as the arguments of the single-run entry point it mirrors),
 * requires all runs to agree in every shape that decides the grid (refused otherwise), and computes for each run exactly what
 * the single-run call computes (bit-identical: tests/test_seed_batch_gpu.py).
 *   args_ws: device scratch of >= n_runs * ICRL_BATCH_ARGS_BYTES bytes, any contents; it receives the per-run argument blocks
 *            (written on `stream` before the launch) and must not be reused by another call before this one has completed. */
#define ICRL_BATCH_ARGS_BYTES 1024

/* icrl_rollout_collect_ex for n_runs runs: ONE persistent launch of grid (n_envs, n_runs) + ONE batched dual-GAE launch (do_gae
 * bit 0).  Batched form exists for the one-workgroup-per-env persistent kernel only (n_envs <= 128, n_envs x obs_dim <= 4096:
 * BASELINE configs[1]); every run needs its exchange workspace (icrl_agent_t.xch_ws) or T * N * 4 >= ICRL_ROLLOUT_WS_BYTES.
 * args_ws: 2 x n_runs x ICRL_BATCH_ARGS_BYTES (rollout + GAE argument blocks). */
typedef struct {
  const icrl_env_t* env;
  const icrl_norm_t* nm;
  const icrl_policy_t* pol;
  const icrl_costnet_t* cn;     /* NULL in every run or in none */
  const icrl_buffer_t* buf;
  const icrl_agent_t* ag;
  const float* noise;           /* [T, N, act] */
} icrl_rollout_job_t;
int icrl_rollout_collect_batch(int n_runs, const icrl_rollout_job_t* jobs, const float* action_low, const float* action_high,
                               double reward_gamma, double reward_gae_lambda, double cost_gamma, double cost_gae_lambda,
                               int do_gae, void* args_ws, long long args_ws_bytes, void* stream);

/* icrl_gae_dual_ws for n_runs [T,N] rollouts of one shape: ONE launch of the two-level scan, grid (tiles * C, n_runs), every run
 * with its own workspace; shapes the split scan does not serve are issued as n_runs single launches. */
typedef struct {
  const float *rewards, *costs, *reward_values, *cost_values, *dones, *last_v_r, *last_v_c;
  const uint8_t* last_dones;
  float *adv_r, *adv_c, *ret_r, *ret_c;
  void* ws;
  long long ws_bytes;
} icrl_gae_job_t;
int icrl_gae_dual_batch(int n_runs, const icrl_gae_job_t* jobs, int T, int N, double reward_gamma, double reward_gae_lambda,
                        double cost_gamma, double cost_gae_lambda, void* args_ws, long long args_ws_bytes, void* stream);

/* icrl_sample_episodes for n_runs runs: grid (n_streams, n_runs); every run has its own env streams, frozen normaliser, policy,
 * noise and outputs; the shape arguments are common. */
typedef struct {
  const icrl_env_t* env;
  const icrl_norm_t* nm;
  const icrl_policy_t* pol;
  const float* noise;
  const int32_t* stream_row0;   /* or NULL (see icrl_sample_episodes) */
  int32_t total_rows, _pad;
  double *orig_obs, *obs;
  float* actions;
  double* ep_rewards;
  int32_t* ep_lengths;
} icrl_sample_job_t;
int icrl_sample_episodes_batch(int n_runs, const icrl_sample_job_t* jobs, const float* action_low, const float* action_high,
                               int episodes_per_stream, int rows_per_stream, int deterministic, int do_reset,
                               void* args_ws, long long args_ws_bytes, void* stream);

/* icrl_cn_train (full-batch mode) for n_runs constraint nets of one architecture: the four launches of an iteration carry all runs
 * (grid.y = run; row counts may differ between runs: grids are sized for the largest, iteration counts likewise). */
typedef struct {
  const icrl_costnet_t* cn;
  float *exp_avg, *exp_avg_sq;
  int32_t* adam_step;
  const float *nominal, *expert;
  int32_t Nn, Ne;
  const int32_t *ep_offsets, *row_episode;
  int32_t n_ep, _pad;
  const icrl_cn_hyper_t* hp;
  float *work, *metrics;
} icrl_cn_train_job_t;
int icrl_cn_train_batch(int n_runs, const icrl_cn_train_job_t* jobs, void* args_ws, long long args_ws_bytes, void* stream);

/* icrl_ppo_lag_train for n_runs runs: 3 | 6 persistent workgroups per run in one grid (layout: "XCD placement" at the top). */
typedef struct {
  const icrl_policy_t* pol;
  float *exp_avg, *exp_avg_sq;
  int32_t* adam_step;
  const icrl_buffer_t* buf;
  const int32_t* perms;
  const float* nu;
  const icrl_ppo_hyper_t* hp;
  float* stats;
  void* sync_ws;
} icrl_ppo_train_job_t;
int icrl_ppo_lag_train_batch(int n_runs, const icrl_ppo_train_job_t* jobs, void* args_ws, long long args_ws_bytes, void* stream);

/* ConstraintNet.prepare_data (icrl/constraint_net.py:258-270): out[n,:] = float32(concat(clip(normalise(obs)), clip(one-hot?
 * acs))[select_dim]).  obs [N,obs] float64, acs [N,acs] float32 (class index when discrete), out [N,in_dim] float32. */
int icrl_cn_prepare(const icrl_costnet_t* cn, const double* obs, const float* acs, int N, float* out, void* stream);

/* ConstraintNet.train after prepare_data (icrl/constraint_net.py:155-229) for the default full-batch mode
 * (batch_size None): per iteration {forward of all nominal + expert rows, compute_is_weights (:231-256: per-episode float32
 * products, KL(old||new), KL(new||old), per-step or per-episode weights incl. the [B,1,1]x[B,1] broadcast of the per-step
 * mode), early-stop test, loss (likelihood-ratio or BCE "gail" form), backward, Adam}.  4 launches per iteration, no host
 * sync; iterations after an early stop are no-ops.
 *   nominal [Nn,in_dim], expert [Ne,in_dim] float32;  ep_offsets [n_ep+1] int32 row offsets of the nominal episodes;
 *   row_episode [Nn] int32;  exp_avg / exp_avg_sq [n_params], adam_step [1] int32 (device): optimiser state;
 *   work: device scratch of icrl_cn_train_work_floats(...) floats;
 *   metrics [iterations][ICRL_CN_METRICS] float32 (device), per iteration i (values of the forward pass made at the START of
 *   iteration i): 0 stop flag, 1 kl_old_new, 2 kl_new_old, 3 is_mean, 4 is_max, 5 is_min, 6 cn_loss, 7 expert_loss,
 *   8 unweighted_nominal_loss, 9 nominal_loss, 10 regularizer_loss, 11..13 nominal preds max/min/mean, 14..16 expert preds
 *   max/min/mean, 17 executed (1 if the optimiser step of this iteration ran). */
size_t icrl_cn_train_work_floats(int n_params, int Nn, int Ne, int n_ep);
int icrl_cn_train(const icrl_costnet_t* cn, float* exp_avg, float* exp_avg_sq, int32_t* adam_step,
                  const float* nominal, const float* expert, int Nn, int Ne,
                  const int32_t* ep_offsets, const int32_t* row_episode, int n_ep,
                  const icrl_cn_hyper_t* hp, float* work, float* metrics, void* stream);

/* Diagnostic: cycles per phase of workgroup 0 of the last persistent rollout that ran with do_gae bit 2 set
 * (out8: policy+env, barrier, tail, steps, exchange load, statistics, normalise, 0). */
int icrl_debug_rollout_profile(unsigned long long* out8);
/* the same for the many-environment persistent kernel: 5 phase sums + T of wave 0 ([0..7]) and of the owner wave ([8..15]) of
 * workgroup 0 (do_gae bit 2) or of the last workgroup, which owns no statistic (do_gae bit 3). */
int icrl_debug_rollout_profile_wide(unsigned long long* out16);
/* per workgroup of that kernel, step T/2 of a profiled launch: 100 MHz timestamps at the end of its env phase, of its owner's
 * gather, of its owner's publish (0 when it owns no statistic) and when it had read all statistics: out[4 * n_workgroups]. */
int icrl_debug_rollout_trace_wide(unsigned long long* out, int n_workgroups);

/* ------------------------------------------------------------------------------------------------------------------
 * Fine-grained pieces of the update, for a host that keeps its MLPs in torch (csrc/fine.hip)
 * ------------------------------------------------------------------------------------------------------------------
 * icrl_ppo_lag_train runs the whole of PPOLagrangian.train() in one launch.  A maintainer who keeps `policy.evaluate_actions` and
 * autograd in torch can bind these instead, one call site at a time (INTEGRATION.md section 5 shows the loop):
 *   rollout_buffer.get(batch_size)         stable_baselines3/common/buffers.py:594-627  -> icrl_minibatch_gather
 *   advantage normalisation                ppo_lag/ppo_lag.py:219-222                   -> icrl_adv_stats
 *   loss terms + their gradients           ppo_lag/ppo_lag.py:224-281                   -> icrl_ppo_lag_loss_fwd_bwd
 *   clip_grad_norm_ + optimizer.step()     ppo_lag/ppo_lag.py:283-288                   -> icrl_clip_adam_step
 *   dual.update_parameter(average_cost)    common/dual_variable.py:47-57                -> icrl_dual_step
 *   rollout_buffer.add(...)                common/buffers.py:554-592                    -> icrl_buffer_add
 *   compute_is_weights / the cn loss       icrl/constraint_net.py:231-256, 188-202      -> icrl_is_weights, icrl_cn_loss_fwd_bwd
 * Launch-bound by construction (a minibatch is 64..512 rows); the fused persistent kernels are the fast path. */

/* flat_idx[n]: env-major indices (i = env * T + t, buffers.py:53-65) -> the rows of the [T, N] buffer: obs [n, obs_dim], actions
 * [n, act_store], and per-row scalars [n]; any output but obs may be NULL. */
int icrl_minibatch_gather(const icrl_buffer_t* buf, const int32_t* flat_idx, int n, float* obs, float* actions, float* old_log_prob,
                          float* adv_r, float* adv_c, float* ret_r, float* ret_c, float* old_v_r, float* old_v_c, void* stream);
/* out4 = {mean(adv_r), 1 / (std(adv_r) + 1e-8) with torch's unbiased std, mean(adv_c), std(adv_r)}; n >= 2. */
int icrl_adv_stats(const float* adv_r, const float* adv_c, int n, float* out4, void* stream);
/* common/utils.py:43-59 `explained_variance(y_pred, y_true)` = 1 - Var[y_true - y_pred] / Var[y_true] (NaN when Var[y_true] == 0), float64
 * sums over n float32 values, for one or two pairs in one pass (the second pair may be NULL as a whole; then out2[1] is not written).
 * Call site replaced: ppo_lag/ppo_lag.py:311-312 — NB the reference passes (returns, values), i.e. y_pred = returns, y_true = values.
 * work: 8 x 256 doubles of device scratch; out2: device float32. */
int icrl_explained_variance(const float* y_pred_a, const float* y_true_a, const float* y_pred_b, const float* y_true_b, long long n,
                            double* work, float* out2, void* stream);
/* The PPO-Lagrangian loss of ONE minibatch on the networks' outputs (all [n] float32; adv_r / adv_c RAW: they are normalised /
 * centred inside as ppo_lag.py:219-222 does): terms8 = {loss, policy_loss, reward_value_loss, cost_value_loss, entropy_loss,
 * approx_kl, clip_fraction, 0}; d_log_prob / d_v_r / d_v_c / d_entropy = d loss / d that output, what loss.backward() would send
 * into `log_prob.backward(...)` etc.  old_v_r / old_v_c NULL: no value clipping; entropy NULL: the reference's -mean(-log_prob)
 * estimate (its gradient then goes into d_log_prob; d_entropy is not written).  nu: device pointer.  2 <= n <= 65536. */
int icrl_ppo_lag_loss_fwd_bwd(const float* log_prob, const float* old_log_prob, const float* adv_r, const float* adv_c, const float* v_r,
                              const float* v_c, const float* ret_r, const float* ret_c, const float* old_v_r, const float* old_v_c,
                              const float* entropy, const float* nu, const icrl_ppo_hyper_t* hp, int n, float* terms8, float* d_log_prob,
                              float* d_v_r, float* d_v_c, float* d_entropy, void* stream);
/* torch.nn.utils.clip_grad_norm_(max_grad_norm) followed by torch.optim.Adam.step() on ONE flat parameter buffer (hp: lr, betas, eps,
 * max_grad_norm; bias corrections in double from *adam_step + 1, which is then incremented on the device).  work: 256 floats of
 * scratch; out2 (may be NULL) = {total norm, clip coefficient}. */
int icrl_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t* adam_step, long long n,
                        const icrl_ppo_hyper_t* hp, float* work, float* out2, void* stream);
/* DualVariable.update_parameter on the device: state4 = {log_nu, exp_avg, exp_avg_sq, nu} (nu = softplus(log_nu) is rewritten, so
 * state4 + 3 can be the `nu` argument of the update), *adam_step its Adam step count; cost from cost_dev[0] when not NULL, else
 * cost_host; clamp_log_nu = the clamp floor in log_nu space (the reference's double inverse softplus, dual_variable.py:19-29);
 * loss_out (may be NULL) = -nu * (cost - budget). */
int icrl_dual_step(float* state4, int32_t* adam_step, const float* cost_dev, float cost_host, float budget, float learning_rate,
                   float clamp_log_nu, float* loss_out, void* stream);

/* RolloutBufferWithCost.add (stable_baselines3/common/buffers.py:554-592): one step's arrays of all N envs (observations float64
 * [N, obs_dim] x 4, action float32 [N, act_store], reward / cost float64 [N], the rest float32 / uint8 [N]) -> row t of the buffer. */
int icrl_buffer_add(const icrl_buffer_t* buf, int t, const double* obs, const double* orig_obs, const double* new_obs, const double* new_orig_obs,
                    const float* action, const double* reward, const double* cost, const float* orig_cost, const uint8_t* done,
                    const float* reward_value, const float* cost_value, const float* log_prob, void* stream);
/* ConstraintNet.compute_is_weights (icrl/constraint_net.py:231-256): preds_old / preds_new [N] = zeta of the start-of-call and of the current
 * network on the nominal rows, episodes = rows ep_offsets[e] .. ep_offsets[e + 1] (n_ep + 1 offsets); weights [N] per step (ratio / mean
 * ratio) or per episode (repeated over its rows); ep_prod [n_ep] scratch / output (float32 products, overflowing like the reference's);
 * out4 = {kl_old_new, kl_new_old, mean ratio, sum of the products}. */
int icrl_is_weights(const float* preds_old, const float* preds_new, int N, const int32_t* ep_offsets, int n_ep, float eps, int per_step,
                    float* weights, float* ep_prod, float* out4, void* stream);
/* The constraint-net loss of one (mini)batch on the network's outputs (icrl/constraint_net.py:188-202): terms6 = {loss, expert_loss,
 * nominal_loss, regularizer, mean(log(nominal + eps)), 0}; d_nominal / d_expert = d loss / d prediction (for preds.backward(...)).
 * is_weights NULL: ones.  mode bit 0: the GAIL discriminator's BCE (gail_utils.py); bit 1: the per-step broadcast quirk of the
 * reference (nominal_loss = mean(w) * mean(log zeta)). */
int icrl_cn_loss_fwd_bwd(const float* nominal_preds, const float* expert_preds, const float* is_weights, int Bn, int Be, float reg_coeff, float eps,
                         int mode, float* terms6, float* d_nominal, float* d_expert, void* stream);

/* Diagnostic (bench.py `roofline.copy_gbs`; no reference counterpart): the memory traffic of the streaming dual-GAE launch without
 * its recurrence.  mode 0: grid, access pattern and bytes of icrl_gae_dual at N >= 131 072 (five [T,N] float arrays read, four written,
 * 16-byte non-temporal accesses, one wave per 256 columns walking the rows downwards); mode 1: in0..in3 copied to out0..out3 by
 * four flat grid-stride float4 copies (in4 unused).  N % 4 == 0. */
int icrl_debug_stream_ref(const float* in0, const float* in1, const float* in2, const float* in3, const float* in4,
                          float* out0, float* out1, float* out2, float* out3, int T, int N, int mode, void* stream);

/* Minibatch mode of ConstraintNet.train (`--cn_batch_size`; icrl/constraint_net.py:181-206 with get() :300-316): per
 * iteration the importance weights / early-stop test on ALL nominal rows as above, then one optimiser step per batch of
 * `batch_size` indices out of perms[iteration] ([iterations][min(Nn,Ne)] int32 on the device, the caller's
 * np.random.permutation stream); a batch takes the same row indices from the nominal and the expert set.  metrics rows hold
 * the importance-sampling statistics of the iteration and the loss terms / predictions of its LAST batch (what the
 * reference reports).  work: icrl_cn_train_work_floats(...) floats as for icrl_cn_train. */
int icrl_cn_train_minibatch(const icrl_costnet_t* cn, float* exp_avg, float* exp_avg_sq, int32_t* adam_step,
                            const float* nominal, const float* expert, int Nn, int Ne,
                            const int32_t* ep_offsets, const int32_t* row_episode, int n_ep,
                            const icrl_cn_hyper_t* hp, const int32_t* perms, int batch_size, float* work,
                            float* metrics, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICRL_HIP_H */
