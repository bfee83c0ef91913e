"""ctypes binding of libuamenv_hip.so (the C-ABI declared in include/uamenv.h).

There is no fallback: if the shared library is missing or no HIP device is
usable, loading / uam_create raises.  Nothing here imports the CPU oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("UAMENV_LIB") or os.path.join(CSRC, "libuamenv_hip.so")   # UAMENV_LIB: development override

OBS_NORM4, OBS_RAW4, OBS_RAW6, OBS_KNN5 = 0, 1, 2, 3
ACT_DISCRETE9, ACT_PAIR, ACT_CONTINUOUS1, ACT_DISCRETE3, ACT_CONTINUOUS2 = 0, 1, 2, 3, 4
DYN_STRAIGHT, DYN_RANDINTRUDER = 0, 1
RNG_PHILOX, RNG_MT19937, RNG_MT19937_SHARED = 0, 1, 2
POS_F32, POS_EXACT = 0, 1
INFO_CODES = ("", "c", "g", "n", "w")   # SingleAircraftEnv.py:168,172,180,183


class UamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("uamenv error %d: %s" % (code, msg))
        self.code = code


class UamConfig(C.Structure):
    """`uam_config` (include/uamenv.h): POD snapshot of Simulators/config.py + batch geometry + modes."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_envs", C.c_int32), ("num_intruders", C.c_int32),
        ("obs_mode", C.c_int32), ("action_mode", C.c_int32), ("dynamics", C.c_int32),
        ("rng_mode", C.c_int32), ("precision", C.c_int32), ("auto_reset", C.c_int32),
        ("sparse_reward", C.c_int32), ("random_own_start", C.c_int32), ("track_episodes", C.c_int32),
        ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
        ("window_width", C.c_double), ("window_height", C.c_double),
        ("minimum_separation", C.c_double), ("nmac_dist", C.c_double),
        ("initial_min_dist", C.c_double), ("goal_radius", C.c_double),
        ("min_speed", C.c_double), ("max_speed", C.c_double), ("d_speed", C.c_double),
        ("speed_sigma", C.c_double), ("position_sigma", C.c_double),
        ("d_heading", C.c_double), ("heading_sigma", C.c_double),
        ("nmac_penalty", C.c_double), ("conflict_penalty", C.c_double),
        ("step_penalty", C.c_double), ("goal_reward", C.c_double), ("dense_reward_div", C.c_double),
        ("own_x0", C.c_double), ("own_y0", C.c_double), ("own_speed0", C.c_double), ("own_heading0", C.c_double),
        ("heading_change_prob", C.c_double), ("heading_change_max_deg", C.c_double),
        ("wall_penalty", C.c_double), ("diagonal", C.c_double), ("knn", C.c_int32), ("wall_terminates", C.c_int32),
        ("conflict_coeff", C.c_double), ("goal_margin", C.c_double), ("shaped_reward", C.c_int32), ("fuse_rollout", C.c_int32),
        ("graph_safe", C.c_int32), ("reserved2", C.c_int32),
    ]


class UamStateView(C.Structure):
    """`uam_state_view` (include/uamenv.h): host pointers for get/set_state."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_envs", C.c_int32), ("num_intruders", C.c_int32), ("reserved0", C.c_int32),
        ("int_x", C.c_void_p), ("int_y", C.c_void_p), ("int_vx", C.c_void_p), ("int_vy", C.c_void_p),
        ("int_speed", C.c_void_p), ("int_heading", C.c_void_p),
        ("int_conflict", C.c_void_p), ("int_promoted", C.c_void_p),
        ("own_x", C.c_void_p), ("own_y", C.c_void_p),
        ("own_heading", C.c_void_p), ("own_speed", C.c_void_p), ("own_vx", C.c_void_p), ("own_vy", C.c_void_p),
        ("goal_x", C.c_void_p), ("goal_y", C.c_void_p),
        ("no_conflict", C.c_void_p), ("reset_count", C.c_void_p),
        ("mt_key", C.c_void_p), ("mt_pos", C.c_void_p), ("mt_has_gauss", C.c_void_p), ("mt_gauss", C.c_void_p),
        ("step_index", C.c_uint64),
    ]


class UamMctsConfig(C.Structure):
    """`uam_mcts_config` (include/uamenv.h): Algorithms/MCTS/config_single.py snapshot + search parameters."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_intruders", C.c_int32), ("simulations", C.c_int32),
        ("search_depth", C.c_int32), ("simulate_frame", C.c_int32), ("model", C.c_int32),
        ("seed", C.c_uint64),
        ("window_width", C.c_double), ("window_height", C.c_double), ("minimum_separation", C.c_double),
        ("min_speed", C.c_double), ("max_speed", C.c_double), ("d_speed", C.c_double),
        ("speed_sigma", C.c_double), ("position_sigma", C.c_double),
        ("d_heading", C.c_double), ("heading_sigma", C.c_double),
        ("heading_change_prob", C.c_double), ("heading_change_max_deg", C.c_double),
        ("c_param", C.c_double), ("reward_div", C.c_double),
    ]


STATE_FIELDS = {
    "int_x": (np.float64, "BN"), "int_y": (np.float64, "BN"),
    "int_vx": (np.float32, "BN"), "int_vy": (np.float32, "BN"),
    "int_speed": (np.float64, "BN"), "int_heading": (np.float64, "BN"),
    "int_conflict": (np.uint8, "BN"), "int_promoted": (np.uint8, "BN"),
    "own_x": (np.float32, "B"), "own_y": (np.float32, "B"),
    "own_heading": (np.float64, "B"), "own_speed": (np.float64, "B"),
    "own_vx": (np.float64, "B"), "own_vy": (np.float64, "B"),
    "goal_x": (np.float64, "B"), "goal_y": (np.float64, "B"),
    "no_conflict": (np.int32, "B"), "reset_count": (np.uint32, "B"),
    "mt_key": (np.uint32, "B624"), "mt_pos": (np.int32, "B"),
    "mt_has_gauss": (np.int32, "B"), "mt_gauss": (np.float64, "B"),
}

# every symbol include/uamenv.h and include/uamenv_bench.h declare: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("uam_config_default", C.c_int, [C.POINTER(UamConfig), C.c_int32, C.c_int32]),
    ("uam_obs_dim", C.c_int32, [C.POINTER(UamConfig)]),
    ("uam_create", C.c_int, [C.POINTER(UamConfig), C.c_int, C.POINTER(_P)]),
    ("uam_destroy", C.c_int, [_P]),
    ("uam_seed", C.c_int, [_P, _P]),
    ("uam_reset", C.c_int, [_P, _P, _P, _P]),
    ("uam_step", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("uam_step_many", C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    ("uam_rollout", C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P]),
    ("uam_rollout_policy", C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint64, _P, C.c_int32, _P, _P, _P, _P, _P]),
    ("uam_steps_per_launch", C.c_int32, [_P, C.c_int32]),
    ("uam_policy_steps_per_launch", C.c_int32, [_P, C.c_int32]),
    ("uam_rollout_kernel", C.c_char_p, [_P, C.c_int32]),
    ("uam_bench_copy", C.c_int, [C.c_int, _P, _P, C.c_int64, _P]),
    ("uam_bench_walk", C.c_int, [C.c_int, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P]),
    ("uam_observe", C.c_int, [_P, _P, _P]),
    ("uam_counters", C.c_int, [_P, _P, _P]),
    ("uam_nearest", C.c_int, [_P, _P, _P]),
    ("uam_her_goal_reward", C.c_int, [C.POINTER(UamConfig), C.c_int, _P, _P, C.c_int64, _P, _P]),
    ("uam_her_input_reward", C.c_int, [C.POINTER(UamConfig), C.c_int, _P, C.c_int64, C.c_int32, _P, _P, _P]),
    ("uam_episode_stats", C.c_int, [_P, _P, _P, _P]),
    ("uam_get_state", C.c_int, [_P, C.POINTER(UamStateView)]),
    ("uam_set_state", C.c_int, [_P, C.POINTER(UamStateView)]),
    ("uam_policy", C.c_int, [_P, C.c_int32, C.c_uint64, _P, _P]),
    ("uam_step_bytes", C.c_int64, [_P]),
    ("uam_rollout_bytes", C.c_int64, [_P, C.c_int32]),
    ("uam_timing_begin", C.c_int, [_P, _P]),
    ("uam_timing_end", C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    ("uam_kernel_timing", C.c_int, [_P, C.c_int32]),
    ("uam_kernel_timing_read", C.c_int, [_P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    ("uam_mcts_config_default", C.c_int, [C.POINTER(UamMctsConfig), C.c_int32]),
    ("uam_mcts_roots_from_env", C.c_int, [_P, _P, _P]),
    ("uam_mcts_workspace_bytes", C.c_int64, [C.POINTER(UamMctsConfig), C.c_int32]),
    ("uam_mcts_best_action", C.c_int, [C.POINTER(UamMctsConfig), C.c_int, _P, C.c_int32, C.c_uint32, _P, _P, _P, _P, _P]),
    ("uam_mcts_move", C.c_int, [C.POINTER(UamMctsConfig), C.c_int, _P, _P, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32,
                                _P, _P, _P, _P]),
    ("uam_last_error", C.c_char_p, []),
    ("uam_version", C.c_char_p, []),
]


def build(force=False):
    """Compile csrc/libuamenv_hip.so for gfx950 with hipcc (works without a GPU)."""
    # make knows every dependency (csrc/Makefile) and is a no-op when the library is up to date
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j2"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no Python/CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `make -C %s` (or __graft_entry__.build()); "
                "uamenv has no CPU fallback" % (LIB_PATH, CSRC))
        # Load order matters in a process that also uses torch: torch ships its own HIP runtime, and whichever
        # libamdhip64 is mapped first is the one both end up talking to.  Loading ours first left uam_create without
        # a visible device once torch came in ("no ROCm-capable device"); torch first works in every order of use.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)   # AttributeError if the library does not export the declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise UamError(rc, lib().uam_last_error().decode("utf-8", "replace"))


def default_config(num_envs, num_intruders, **over):
    cfg = UamConfig()
    check(lib().uam_config_default(C.byref(cfg), num_envs, num_intruders))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError("uam_config has no field %r" % k)
        setattr(cfg, k, v)
    return cfg


def obs_dim(cfg):
    return int(lib().uam_obs_dim(C.byref(cfg)))


def make_view(arrays, B, N):
    v = UamStateView()
    v.struct_size = C.sizeof(UamStateView)
    v.num_envs, v.num_intruders = B, N
    v.step_index = 0xFFFFFFFFFFFFFFFF   # uam_set_state: keep the handle's step index unless given
    for name, arr in arrays.items():
        if name == "step_index":
            v.step_index = int(arr)
            continue
        if name not in STATE_FIELDS:
            raise KeyError(name)
        dt, shape = STATE_FIELDS[name]
        want = {"B": (B,), "BN": (B, N), "B624": (B, 624)}[shape]
        if arr.dtype != dt or tuple(arr.shape) != want or not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("state field %s must be C-contiguous %s%s, got %s%s" %
                             (name, np.dtype(dt).name, want, arr.dtype, arr.shape))
        setattr(v, name, arr.ctypes.data)
    return v
