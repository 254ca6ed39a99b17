"""ctypes binding of csrc/libcovo_hip.so (C ABI: include/covo_hip.h).

Fails loudly: a missing library is an ImportError-class failure at first use, never a silent
fallback to another backend.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("COVO_HIP_LIB") or os.path.join(_HERE, "csrc", "libcovo_hip.so")  # COVO_HIP_LIB: A/B builds
_lib = None

COVO_H = 32
COVO_DU = 4
COVO_NA = COVO_H * COVO_DU
COVO_STATE_FLOATS = 32
COVO_PARTIAL_FLOATS = 132
COVO_POS_STATS_DOUBLES = COVO_H * 6
COVO_RANK_RECORD_FLOATS = COVO_PARTIAL_FLOATS + 2 * COVO_POS_STATS_DOUBLES  # 516: {m, s, v[128], pad} + 192 fp64 position sums
COVO_COV_FLOATS = COVO_H * 10
COVO_RANK_RECORD_COV_FLOATS = COVO_PARTIAL_FLOATS + COVO_COV_FLOATS + 2 * COVO_POS_STATS_DOUBLES  # 836: with MPPI's second moments
COVO_EXCHANGE_HANDLE_BYTES = 128
ABI_VERSION = 7
COVO_FLAG_ACTIONS_CLIPPED = 1


class CovoError(RuntimeError):
    pass


class EnvParamsC(C.Structure):
    """struct covo_env_params (include/covo_hip.h)."""
    _fields_ = [
        ("max_thrust", C.c_float), ("max_torque", C.c_float * 3), ("max_omega", C.c_float * 3),
        ("dt", C.c_float), ("g", C.c_float), ("m", C.c_float), ("action_scale", C.c_float),
        ("alpha_bodyrate", C.c_float), ("max_steps_in_episode", C.c_int32), ("pos_limit", C.c_float),
        ("rollover_terminate", C.c_int32), ("reward_kind", C.c_int32), ("disturb_kind", C.c_int32),
        ("disturb_period", C.c_int32), ("disturb_scale", C.c_float), ("disturb_params", C.c_float * 6),
        ("dyn_noise_scale", C.c_float),
        ("reset_traj", C.c_int32), ("reserved0", C.c_int32), ("reset_dt", C.c_double), ("reset_disturb_scale", C.c_double),
    ]


class ConfigC(C.Structure):
    """struct covo_config (include/covo_hip.h)."""
    _fields_ = [("n_local", C.c_int32), ("H", C.c_int32), ("du", C.c_int32), ("lam", C.c_float),
                ("discount", C.c_float), ("flags", C.c_int32)]


_P = C.c_void_p


class StepArgsC(C.Structure):
    """struct covo_step_args (include/covo_hip.h)."""
    _fields_ = [("mode", C.c_int32), ("n_samples", C.c_int32), ("T", C.c_int32), ("n_table", C.c_int32),
                ("state", _P), ("pos_traj", _P), ("vel_traj", _P), ("a_mean", _P), ("a_mean_shift", _P), ("a_cov", _P),
                ("L_table", _P), ("a", _P), ("cost", _P), ("groupmin", _P), ("pos_stats", _P), ("partial_out", _P),
                ("sample_offset", C.c_int64), ("gamma_mean", C.c_float), ("sample_sigma", C.c_float),
                ("derive_keys", C.c_int32), ("rollout_deterministic", C.c_int32), ("gamma_sigma", C.c_float),
                ("pad_", C.c_int32), ("a_mean_in", _P)]


class BatchArgsC(C.Structure):
    """struct covo_batch_args (include/covo_hip.h)."""
    _fields_ = [("n_envs", C.c_int32), ("n_samples", C.c_int32), ("T", C.c_int32), ("pad_", C.c_int32),
                ("states", _P), ("pos_traj", _P), ("vel_traj", _P), ("a_mean", _P), ("a_cov", _P), ("a", _P), ("cost", _P),
                ("groupmin", _P), ("gamma_mean", C.c_float), ("sample_sigma", C.c_float)]


REWARD_KINDS = {"penyaw": 0, "realworld": 1}                 # COVO_REWARD_*
DISTURB_KINDS = {"none": 0, "gaussian": 1, "periodic": 2, "sin": 3, "drag": 4, "mixed": 5}  # COVO_DISTURB_*
TRAJ_KINDS = {None: 0, "hovering": 1, "tracking": 2, "tracking_slow": 3, "tracking_zigzag": 4}  # COVO_TRAJ_* by Quad3D task
TABLE_DISTURB_KINDS = (2, 3, 4, 5)                            # models that need covo_disturb_table's per-step table
DISTURB_KEYS_SHARED, DISTURB_KEYS_HESSIAN, DISTURB_KEYS_NOMINAL = 0, 1, 2
COVO_MAX_ENVS = 64
MODE_MPPI, MODE_COVO_ONLINE, MODE_COVO_OFFLINE = 0, 1, 2
COVO_FLAG_NO_GRAPH = 2
COVO_FLAG_SHARED_DEVICE = 4
COVO_FLAG_PROPAGATE_NAN = 8
COVO_E_DEVICE = -4
COVO_DEVSTAT_GRID_BARRIER = 1
COVO_DEVSTAT_EXCHANGE = 2
COVO_DEVSTAT_ADJOINT = 4
_SIGS = {
    "covo_last_error": (C.c_char_p, []),
    "covo_abi_version": (C.c_int, []),
    "covo_create": (C.c_int, [C.POINTER(ConfigC), C.POINTER(_P)]),
    "covo_destroy": (C.c_int, [_P]),
    "covo_device_status": (C.c_int, [_P, C.c_int32]),
    "covo_debug_raise_device_status": (C.c_int, [_P, C.c_int32, _P]),
    "covo_randn": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "covo_randn_jax": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "covo_noise_gemm": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P]),
    "covo_noise_blockdiag": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P]),
    "covo_noise_gemm_philox": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, C.c_int64, C.c_int32, _P, _P]),
    "covo_noise_blockdiag_philox": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, C.c_int64, C.c_int32, _P, _P]),
    "covo_rollout_cost": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), C.POINTER(C.c_float), _P, _P,
                                    C.c_int32, _P, _P, _P, _P]),
    "covo_disturb_table": (C.c_int, [_P, C.POINTER(EnvParamsC), _P, C.c_int32, _P, C.c_uint32, C.c_uint32, C.c_int32,
                                     C.c_int32, _P, _P]),
    "covo_pos_info": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P]),
    "covo_debug_time_rollout": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), C.POINTER(C.c_float), _P, _P,
                                          C.c_int32, _P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_float), _P]),
    "covo_softmax_reduce": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P]),
    "covo_softmax_update": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_float, _P, _P]),
    "covo_softmax_update_cov": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_float, _P, C.c_float, _P, _P, _P]),
    "covo_merge": (C.c_int, [_P, _P, C.c_int32, _P, C.c_float, _P, _P]),
    "covo_merge_ranks": (C.c_int, [_P, _P, C.c_int32, _P, C.c_float, _P, _P, _P]),
    "covo_merge_ranks_wide": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_float, _P, _P, _P]),
    "covo_exchange_create": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "covo_exchange_connect": (C.c_int, [_P, _P]),
    "covo_exchange_set_timeout": (C.c_int, [_P, C.c_double]),
    "covo_device_bus_id": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "covo_exchange_records": (C.c_int, [_P, _P, _P, _P]),
    "covo_exchange_records_cov": (C.c_int, [_P, _P, _P, _P]),
    "covo_softmax_reduce_cov": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "covo_merge_ranks_cov": (C.c_int, [_P, _P, C.c_int32, _P, C.c_float, _P, C.c_float, _P, _P, _P, _P]),
    "covo_shift_mean": (C.c_int, [_P, _P, _P, _P]),
    "covo_hessian": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), _P, _P, C.c_int32, _P, _P]),
    "covo_hessian_pairs": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), _P, _P, C.c_int32, _P, _P]),
    "covo_sigma": (C.c_int, [_P, _P, C.c_int32, C.c_float, _P, _P, _P]),
    "covo_debug_sigma_workspace": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "covo_debug_hess_workspace": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "covo_debug_batched_hessians": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "covo_env_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), _P, C.POINTER(C.c_uint32),
                                C.c_int32, C.c_float, _P, C.c_int32, _P]),
    "covo_pid_nominal": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), C.POINTER(EnvParamsC), C.c_float,
                                   C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_int32, _P, _P, _P, _P]),
    "covo_run_episode": (C.c_int, [_P, C.POINTER(EnvParamsC), C.POINTER(StepArgsC), _P, _P, C.c_int32,
                                   C.c_float, _P, C.POINTER(C.c_uint32), C.c_int32, _P]),
    "covo_mpc_step_batched": (C.c_int, [_P, C.POINTER(BatchArgsC), C.POINTER(EnvParamsC), C.POINTER(C.c_uint32), _P]),
    "covo_env_step_batched": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.POINTER(EnvParamsC), _P,
                                        C.POINTER(C.c_uint32), C.c_int32, C.c_float, _P, C.c_int32, C.c_int32, _P]),
    "covo_run_episode_batched": (C.c_int, [_P, C.POINTER(BatchArgsC), C.POINTER(EnvParamsC), _P, _P, C.c_int32, C.c_float, _P,
                                           C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, _P]),
    "covo_debug_set_ns_tail": (C.c_int, [_P, C.c_int, C.c_int]),       # per-handle experiment switches (covo_hip.h)
    "covo_debug_set_ns_deflate": (C.c_int, [_P, C.c_int]),
    "covo_debug_set_ns_ritz_inside": (C.c_int, [_P, C.c_int]),
    "covo_debug_set_fuse_small": (C.c_int, [_P, C.c_int]),
    "covo_debug_set_fold_begin": (C.c_int, [_P, C.c_int]),
    "covo_debug_set_stream_gemm": (C.c_int, [_P, C.c_int]),
    "covo_debug_set_ns_coherence": (C.c_int, [_P, C.c_int]),
    "covo_debug_set_ns_merged": (C.c_int, [_P, C.c_int]),
    "covo_debug_time_step": (C.c_int, [_P, C.POINTER(EnvParamsC), C.POINTER(StepArgsC), C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.POINTER(C.c_float), _P]),
    "covo_debug_time_batched": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_float), _P]),
    "covo_sigma_jacobi": (C.c_int, [_P, _P, C.c_int32, C.c_float, _P, _P, _P]),
    "covo_sigma_profile": (C.c_int, [_P, _P, C.c_float, _P, _P, _P, _P]),
    "covo_mpc_step": (C.c_int, [_P, C.POINTER(EnvParamsC), C.POINTER(StepArgsC), C.c_uint32, C.c_uint32,
                                C.POINTER(C.c_float), _P]),
    "covo_cholesky": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
}
EXPORTS = tuple(_SIGS)


def lib_path() -> str:
    return _SO


def load_library():
    """Load libcovo_hip.so and declare every prototype of include/covo_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise CovoError(
            f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C covo_mpc_amd/csrc`).  covo_mpc_amd has no CPU/torch fallback.")
    # torch bundles its own ROCm runtime (libamdhip64): load it FIRST so this library binds to the
    # runtime instance that owns torch's device context and streams (two runtimes in one process do
    # not see each other's devices).
    import torch  # noqa: F401
    lib = C.CDLL(_SO)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, surfaced as is
        fn.restype = res
        fn.argtypes = args
    if lib.covo_abi_version() != ABI_VERSION:
        raise CovoError(f"libcovo_hip.so ABI {lib.covo_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().covo_last_error()
        raise CovoError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
