"""ctypes binding of libicrl_hip.so (the C ABI declared in include/icrl_hip.h).

The library is the product path: if it is missing this module raises at import of the first symbol —
there is no CPU / PyTorch fallback anywhere in ``icrl_amd``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICRL_HIP_LIB") or os.path.join(_HERE, "lib", "libicrl_hip.so")      # ICRL_HIP_LIB: a variant build (A/B timing)

_lib = None
PPO_SPLIT_BYTES = 2560 * 1024      # ICRL_PPO_SPLIT_BYTES

c_void_p, c_int, c_double, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_float

# name -> argtypes (restype is always int = hipError_t).  Kept in one table so that the CPU test-suite can check that
# every symbol declared in include/icrl_hip.h is exported and bound.
SIGNATURES = {
    "icrl_abi_version": [],
    "icrl_last_error": [],
    "icrl_clear_error": [],
    "icrl_gae_dual": [c_void_p] * 12 + [c_int, c_int] + [c_double] * 4 + [c_void_p],
    "icrl_gae_dual_ex": [c_void_p] * 12 + [c_int, c_int] + [c_double] * 4 + [c_int, c_void_p],
    "icrl_gae_dual_ws": [c_void_p] * 12 + [c_int, c_int] + [c_double] * 4 + [c_int, c_void_p, ctypes.c_longlong, c_void_p],
    "icrl_gae_dual_ws_bytes": [c_int, c_int],
    "icrl_policy_prepare": [c_void_p, c_void_p],
    "icrl_costnet_prepare": [c_void_p, c_void_p],
    "icrl_policy_forward": [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 8,
    "icrl_policy_evaluate": [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 5,
    "icrl_sample_episodes": [c_void_p] * 6 + [c_int] * 4 + [c_void_p, c_int] + [c_void_p] * 6,
    "icrl_cost_mlp_forward": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "icrl_disc_reward": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "icrl_synth_env_reset": [c_void_p, c_void_p],
    "icrl_synth_env_step": [c_void_p] * 5,
    "icrl_vecnorm_reset": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "icrl_vecnorm_step": [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 4,
    "icrl_rollout_collect": [c_void_p] * 9 + [c_double] * 4 + [c_void_p],
    "icrl_rollout_collect_ex": [c_void_p] * 9 + [c_double] * 4 + [c_int, c_void_p],
    "icrl_ppo_lag_train": [c_void_p] * 11,
    "icrl_ppo_generic_row_floats": [c_void_p],
    "icrl_cn_prepare": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "icrl_cn_train_work_floats": [c_int, c_int, c_int, c_int],
    "icrl_cn_train": [c_void_p] * 6 + [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "icrl_debug_rollout_profile": [c_void_p],
    "icrl_debug_rollout_profile_wide": [c_void_p],
    "icrl_debug_rollout_trace_wide": [c_void_p, c_int],
    "icrl_debug_stream_ref": [c_void_p] * 9 + [c_int, c_int, c_int, c_void_p],
    # fine-grained pieces of the update for a host that keeps torch MLPs (csrc/fine.hip)
    "icrl_minibatch_gather": [c_void_p, c_void_p, c_int] + [c_void_p] * 10,
    "icrl_adv_stats": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "icrl_ppo_lag_loss_fwd_bwd": [c_void_p] * 13 + [c_int] + [c_void_p] * 6,
    "icrl_clip_adam_step": [c_void_p] * 5 + [ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_void_p],
    "icrl_explained_variance": [c_void_p] * 4 + [ctypes.c_longlong, c_void_p, c_void_p, c_void_p],
    "icrl_dual_step": [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p],
    "icrl_buffer_add": [c_void_p, c_int] + [c_void_p] * 13,
    "icrl_is_weights": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "icrl_cn_loss_fwd_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "icrl_cn_train_minibatch": [c_void_p] * 6 + [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p, c_void_p],
    # batched forms (several independent runs in one launch, run = blockIdx.y): n_runs, jobs[n_runs], ..., args_ws, bytes, stream
    "icrl_rollout_collect_batch": [c_int, c_void_p, c_void_p, c_void_p] + [c_double] * 4 + [c_int, c_void_p, ctypes.c_longlong, c_void_p],
    "icrl_gae_dual_batch": [c_int, c_void_p, c_int, c_int] + [c_double] * 4 + [c_void_p, ctypes.c_longlong, c_void_p],
    "icrl_sample_episodes_batch": [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, ctypes.c_longlong, c_void_p],
    "icrl_cn_train_batch": [c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p],
    "icrl_ppo_lag_train_batch": [c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p],
}
BATCH_ARGS_BYTES = 1024      # ICRL_BATCH_ARGS_BYTES
RESTYPES = {"icrl_cn_train_work_floats": ctypes.c_size_t, "icrl_gae_dual_ws_bytes": ctypes.c_size_t, "icrl_last_error": ctypes.c_char_p, "icrl_clear_error": None}


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; torch must be imported first so that the process-wide HIP runtime
    (libamdhip64.so.7 shipped inside the torch wheel) is the one the library binds to."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (loads libamdhip64 with RTLD_GLOBAL semantics for the soname lookup)
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C icrl_amd/csrc`). icrl_amd has no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, c_int)
    return _lib


byref = ctypes.byref


def check(err, what):
    """hipError_t -> exception.  Refused arguments (hipErrorInvalidValue = 1) carry the library's reason and raise ValueError, as
    the reference's argument checks do; anything else is a runtime failure of the launch."""
    if err != 0:
        reason = lib().icrl_last_error().decode() if err == 1 else ""
        if reason:
            lib().icrl_clear_error()
            raise ValueError(f"{what}: {reason}")
        raise RuntimeError(f"{what} failed with hipError_t {err}")


def ptr(t):
    """data pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
