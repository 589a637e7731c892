"""ctypes mirrors of the plain-C descriptors in include/icrl_hip.h (host structs holding device pointers)."""
import ctypes as C

i32, f64, vp = C.c_int32, C.c_double, C.c_void_p


class EnvT(C.Structure):
    _fields_ = [("n_envs", i32), ("obs_dim", i32), ("act_dim", i32), ("max_steps", i32), ("reward_form", i32),
                ("wall_terminate", i32), ("broken", i32), ("_pad", i32),
                ("B", vp), ("s", vp), ("t_ep", vp), ("step_count", vp), ("key", vp)]


class NormT(C.Structure):
    _fields_ = [("training", i32), ("norm_obs", i32), ("norm_reward", i32), ("norm_cost", i32),
                ("clip_obs", f64), ("clip_reward", f64), ("clip_cost", f64), ("reward_gamma", f64), ("cost_gamma", f64),
                ("epsilon", f64),
                ("obs_mean", vp), ("obs_var", vp), ("obs_count", vp), ("ret_stats", vp), ("cost_stats", vp),
                ("ret", vp), ("cost_ret", vp)]


class PolicyT(C.Structure):
    _fields_ = [("obs_dim", i32), ("act_dim", i32), ("h1", i32), ("h2", i32), ("discrete", i32), ("n_params", i32),
                ("params", vp), ("params_t", vp), ("arch", vp)]      # arch: host int32 array or NULL (include/icrl_hip.h)


class CostNetT(C.Structure):
    _fields_ = [("obs_dim", i32), ("acs_dim", i32), ("in_dim", i32), ("n_hidden", i32), ("h1", i32), ("h2", i32), ("h3", i32), ("h4", i32),
                ("is_discrete", i32), ("n_params", i32),
                ("clip_obs", f64), ("select_dim", vp), ("action_low", vp), ("action_high", vp), ("obs_mean", vp),
                ("obs_var", vp), ("eps", f64), ("params", vp), ("params_t", vp)]


class BufferT(C.Structure):
    _fields_ = [("T", i32), ("N", i32), ("obs_dim", i32), ("act_store", i32),
                ("observations", vp), ("new_observations", vp), ("orig_observations", vp), ("new_orig_observations", vp),
                ("actions", vp), ("dones", vp), ("log_probs", vp), ("rewards", vp), ("reward_values", vp), ("costs", vp),
                ("orig_costs", vp), ("cost_values", vp), ("reward_advantages", vp), ("reward_returns", vp),
                ("cost_advantages", vp), ("cost_returns", vp), ("gae_ws", vp), ("gae_ws_bytes", C.c_longlong)]


class AgentT(C.Structure):
    _fields_ = [("last_obs", vp), ("last_dones", vp), ("raw_rew", vp), ("raw_cost", vp), ("dones", vp),
                ("last_v_r", vp), ("last_v_c", vp), ("act_clipped", vp), ("status", vp), ("xch_ws", vp), ("xch_ws_bytes", C.c_longlong)]


class PpoHyperT(C.Structure):
    _fields_ = [("batch_size", i32), ("n_epochs", i32), ("use_target_kl", i32), ("_pad", i32),
                ("clip_range", C.c_float), ("ent_coef", C.c_float), ("reward_vf_coef", C.c_float), ("cost_vf_coef", C.c_float),
                ("max_grad_norm", C.c_float), ("target_kl", C.c_float), ("clip_range_reward_vf", C.c_float),
                ("clip_range_cost_vf", C.c_float), ("lr", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float),
                ("adam_eps", C.c_float)]


class CnHyperT(C.Structure):
    _fields_ = [("iterations", i32), ("importance_sampling", i32), ("per_step", i32), ("gail", i32),
                ("reg_coeff", C.c_float), ("eps", C.c_float), ("target_kl_old_new", C.c_float), ("target_kl_new_old", C.c_float),
                ("lr", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float)]


# ---- job descriptors of the batched entry points: pointers to the per-run descriptors above (host memory) + device pointers
class RolloutJobT(C.Structure):
    _fields_ = [("env", vp), ("nm", vp), ("pol", vp), ("cn", vp), ("buf", vp), ("ag", vp), ("noise", vp)]


class GaeJobT(C.Structure):
    _fields_ = [(k, vp) for k in ("rewards", "costs", "reward_values", "cost_values", "dones", "last_v_r", "last_v_c", "last_dones",
                                  "adv_r", "adv_c", "ret_r", "ret_c", "ws")] + [("ws_bytes", C.c_longlong)]


class SampleJobT(C.Structure):
    _fields_ = ([(k, vp) for k in ("env", "nm", "pol", "noise", "stream_row0")] + [("total_rows", i32), ("_pad", i32)] +
                [(k, vp) for k in ("orig_obs", "obs", "actions", "ep_rewards", "ep_lengths")])


class CnTrainJobT(C.Structure):
    _fields_ = [("cn", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("adam_step", vp), ("nominal", vp), ("expert", vp), ("Nn", i32), ("Ne", i32),
                ("ep_offsets", vp), ("row_episode", vp), ("n_ep", i32), ("_pad", i32), ("hp", vp), ("work", vp), ("metrics", vp)]


class PpoTrainJobT(C.Structure):
    _fields_ = [(k, vp) for k in ("pol", "exp_avg", "exp_avg_sq", "adam_step", "buf", "perms", "nu", "hp", "stats", "sync_ws")]


def addr(struct):
    """host address of a ctypes struct (for the pointer fields of a job descriptor); the struct must outlive the call."""
    return C.addressof(struct) if struct is not None else None


def p(t):
    """device pointer of a contiguous tensor (or None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return t.data_ptr()

GAE_WS_BYTES = 256 * (256 * 8 + 4) + 64       # ICRL_GAE_WS_BYTES
