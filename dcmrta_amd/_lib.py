"""ctypes loader of the HIP product library (dcmrta_amd/libdcmrta_hip.so).

There is deliberately no fallback: if the library is missing or cannot be loaded the import
of the binding raises, and dcm_create() itself refuses to run without a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCMRTA_HIP_LIB: developer override used by tools/variants.py to A/B differently-compiled builds of the same sources
LIB_PATH = os.environ.get("DCMRTA_HIP_LIB") or os.path.join(_HERE, "libdcmrta_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dcmrta_env.h")

ABI_VERSION = 5
FOLLOWER_COLS = 4
MAX_MEMBERS = 5
MAX_MEMBERS_WIDE = 16    # DCM_PARAM_WIDE_MEMBERS handles
PARAM_NO_GROUPING, PARAM_AUTO_RESET, PARAM_STRICT_MASK, PARAM_WIDE_MEMBERS = 1, 2, 4, 8
MAX_AGENTS = 128
MAX_TASKS = 1023

FLAG_DONE, FLAG_FINISHED, FLAG_TRUNCATED, FLAG_BAD_ACTION, FLAG_OVERFLOW, FLAG_BAD_LEADER, FLAG_TYPE_ERROR = 1, 2, 4, 8, 16, 32, 64
FLAG_BAD_INSTANCE = 256  # a requirement outside 1..MAX_MEMBERS reached dcm_load_instances: the env never starts
FLAG_WAIT_ORDER = 128   # informational: a per-(agent, task) abandonment counter saturated (RL mode) / replay log overflow


class DcmParams(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_agents", C.c_int32), ("n_tasks", C.c_int32), ("device", C.c_int32),
                ("max_waiting_time", C.c_double), ("max_time", C.c_double), ("flags", C.c_uint32),
                ("auto_reset_episodes", C.c_uint32)]


class DcmError(RuntimeError):
    pass


_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
# name -> (restype, argtypes): every symbol include/dcmrta_env.h declares
SIGNATURES = {
    "dcm_last_error": (C.c_char_p, []),
    "dcm_abi_version": (C.c_int, []),
    "dcm_build_id": (C.c_char_p, []),
    "dcm_create": (C.c_int, [C.POINTER(DcmParams), C.POINTER(_vp)]),
    "dcm_destroy": (C.c_int, [_vp]),
    "dcm_load_instances": (C.c_int, [_vp] * 6),
    "dcm_load_instances_ragged": (C.c_int, [_vp] * 8),   # env, depot, task_xy, req, dur, n_agents (host), n_tasks (host), stream
    "dcm_reset": (C.c_int, [_vp] * 3),
    "dcm_observe": (C.c_int, [_vp] * 8),
    "dcm_step": (C.c_int, [_vp] * 11),
    "dcm_rollout_random": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),   # env, episodes, max_decisions, max_decisions_in, obs x3, steps, stream
    "dcm_summary": (C.c_int, [_vp] * 3),
    "dcm_env_status": (C.c_int, [_vp] * 5),
    "dcm_get_tasks": (C.c_int, [_vp] * 10),
    "dcm_get_agents": (C.c_int, [_vp] * 12),
    "dcm_get_abandoned": (C.c_int, [_vp] * 3),
    "dcm_env_episodes": (C.c_int, [_vp] * 3),
    "dcm_get_members": (C.c_int, [_vp] * 3),
    "dcm_state_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "dcm_clone_state": (C.c_int, [_vp] * 3),
    "dcm_restore_state": (C.c_int, [_vp] * 3),
    "dcm_distance": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dcm_record_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "dcm_set_route_log": (C.c_int, [_vp, _vp, _vp, _vp, _i32]),
    "dcm_load_routes": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "dcm_execute_routes": (C.c_int, [_vp, _i32] + [_vp] * 11),
    "dcm_set_visibility": (C.c_int, [_vp, _i32, _i32, _i32, _i32]),
    "dcm_set_replay_placement": (C.c_int, [_vp, _i32]),
    "dcm_set_return_log": (C.c_int, [_vp, _vp, _i32]),
}

_LIB = None


def load():
    """Load libdcmrta_hip.so and bind every entry point. Raises if the library is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DcmError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64; load it FIRST so that this library binds to the same HIP runtime instance
        # the tensors come from (loaded the other way round the process ends up with two runtimes and dcm_create sees no device)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.dcm_abi_version() != ABI_VERSION:
            raise DcmError("libdcmrta_hip.so ABI version mismatch")
        _LIB = lib
    return _LIB


def build_id():
    """sha256 prefix of the kernel sources + flags the loaded library was built from (dcm_build_id)."""
    return load().dcm_build_id().decode()


def check(rc):
    if rc != 0:
        msg = load().dcm_last_error()
        raise DcmError(f"dcmrta_hip error {rc}: {msg.decode() if msg else '?'}")
