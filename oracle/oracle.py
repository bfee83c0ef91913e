"""ctypes front-end of the CPU oracle (oracle/uam_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

OBS_NORM4, OBS_RAW4, OBS_RAW6, OBS_KNN5 = 0, 1, 2, 3
ACT_DISCRETE9, ACT_PAIR, ACT_CONTINUOUS1, ACT_DISCRETE3, ACT_CONTINUOUS2 = 0, 1, 2, 3, 4
DYN_STRAIGHT, DYN_RANDINTRUDER = 0, 1
RNG_PHILOX, RNG_MT19937, RNG_MT19937_SHARED = 0, 1, 2
POS_F32, POS_EXACT = 0, 1
MATH_LIBM, MATH_PORTABLE = 0, 1
INFO_CODES = ["", "c", "g", "n", "w"]


class UamConfig(C.Structure):
    """Mirror of `uam_config` (include/uamenv.h)."""
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
    """Mirror of `uam_state_view` (include/uamenv.h)."""
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
    """Mirror of `uam_mcts_config` (include/uamenv.h)."""
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
    # name: (dtype, per-intruder?)
    "int_x": (np.float64, True), "int_y": (np.float64, True),
    "int_vx": (np.float32, True), "int_vy": (np.float32, True),
    "int_speed": (np.float64, True), "int_heading": (np.float64, True),
    "int_conflict": (np.uint8, True), "int_promoted": (np.uint8, True),
    "own_x": (np.float32, False), "own_y": (np.float32, False),
    "own_heading": (np.float64, False), "own_speed": (np.float64, False),
    "own_vx": (np.float64, False), "own_vy": (np.float64, False),
    "goal_x": (np.float64, False), "goal_y": (np.float64, False),
    "no_conflict": (np.int32, False), "reset_count": (np.uint32, False),
}
MT_FIELDS = {"mt_key": (np.uint32, 624), "mt_pos": (np.int32, 1), "mt_has_gauss": (np.int32, 1),
             "mt_gauss": (np.float64, 1)}


def alloc_state(B, N, with_mt=False):
    """Dict of zeroed host arrays for every uam_state_view field."""
    st = {}
    for name, (dt, per_int) in STATE_FIELDS.items():
        st[name] = np.zeros((B, N) if per_int else (B,), dtype=dt)
    if with_mt:
        for name, (dt, w) in MT_FIELDS.items():
            st[name] = np.zeros((B, w) if w > 1 else (B,), dtype=dt)
    return st


def make_view(st, B, N):
    v = UamStateView()
    v.struct_size = C.sizeof(UamStateView)
    v.num_envs, v.num_intruders = B, N
    v.step_index = 0xFFFFFFFFFFFFFFFF   # keep
    for name, arr in st.items():
        if name == "step_index":
            v.step_index = int(arr)
            continue
        assert arr.flags["C_CONTIGUOUS"]
        setattr(v, name, arr.ctypes.data)
    return v


def build(force=False):
    """Compile oracle/_build/liboracle.so with the committed Makefile (gcc)."""
    # make knows every dependency (oracle/Makefile: the .c, its header and the product's portable-math headers)
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(UamConfig), C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_seed.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.orc_observe.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_state.argtypes = [C.c_void_p, C.POINTER(UamStateView)]
        L.orc_set_state.argtypes = [C.c_void_p, C.POINTER(UamStateView)]
        L.orc_episode_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_her_goal_reward.restype = C.c_double
        L.orc_her_goal_reward.argtypes = [C.POINTER(UamConfig), C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_her_input_reward.restype = C.c_double
        L.orc_her_input_reward.argtypes = [C.POINTER(UamConfig), C.c_int, C.c_void_p, C.c_int]
        L.orc_spawn_count.restype = C.c_int64
        L.orc_spawn_count.argtypes = [C.c_void_p]
        L.orc_redraw_count.restype = C.c_int64
        L.orc_redraw_count.argtypes = [C.c_void_p]
        L.uam_config_default.argtypes = [C.POINTER(UamConfig), C.c_int32, C.c_int32]
        L.uam_obs_dim.argtypes = [C.POINTER(UamConfig)]
        L.uam_obs_dim.restype = C.c_int32
        L.orc_np_seed_random_sample.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.orc_np_seed_normal.argtypes = [C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.orc_np_seed_state.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pmath_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_pmath_log.argtypes = [C.c_double]
        L.orc_pmath_log.restype = C.c_double
        L.uam_mcts_config_default.argtypes = [C.POINTER(UamMctsConfig), C.c_int32]
        L.orc_mcts_move.argtypes = [C.POINTER(UamMctsConfig), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_mcts_best_action.argtypes = [C.POINTER(UamMctsConfig), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def default_config(B, N, **over):
    cfg = UamConfig()
    lib().uam_config_default(C.byref(cfg), B, N)
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data


class OracleBatch:
    """B reference-faithful envs on the CPU; float64 observations like the reference."""

    def __init__(self, cfg, math_impl=MATH_LIBM, shared_stream=False, threads=1):
        self.cfg = cfg
        self.B, self.N = cfg.num_envs, cfg.num_intruders
        self.D = lib().uam_obs_dim(C.byref(cfg))
        self._h = lib().orc_create(C.byref(cfg), math_impl, int(shared_stream))
        if not self._h:
            raise ValueError("orc_create failed (bad config)")
        if threads > 1:
            lib().orc_set_threads(self._h, threads)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    __del__ = close

    def seed(self, seeds=None):
        arr = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        lib().orc_seed(self._h, _ptr(arr))

    def reset(self, mask=None, want_obs=True):
        obs = np.zeros((self.B, self.D), dtype=np.float64) if want_obs else None
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_reset(self._h, _ptr(m), _ptr(obs))
        return obs

    def _actions(self, actions):
        am = self.cfg.action_mode
        if am == ACT_CONTINUOUS1:
            return np.ascontiguousarray(actions, dtype=np.float32).reshape(self.B)
        if am == ACT_CONTINUOUS2:
            return np.ascontiguousarray(actions, dtype=np.float32).reshape(self.B, 2)
        if am == ACT_PAIR:
            return np.ascontiguousarray(actions, dtype=np.int32).reshape(self.B, 2)
        return np.ascontiguousarray(actions, dtype=np.int32).reshape(self.B)

    def step(self, actions, want_obs=True):
        a = self._actions(actions)
        obs = np.zeros((self.B, self.D), dtype=np.float64) if want_obs else None
        rew = np.zeros(self.B, dtype=np.float64)
        done = np.zeros(self.B, dtype=np.uint8)
        info = np.zeros(self.B, dtype=np.uint8)
        lib().orc_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(info))
        return obs, rew, done, info

    def observe(self):
        obs = np.zeros((self.B, self.D), dtype=np.float64)
        lib().orc_observe(self._h, _ptr(obs))
        return obs

    def get_state(self):
        with_mt = self.cfg.rng_mode == RNG_MT19937
        st = alloc_state(self.B, self.N, with_mt)
        v = make_view(st, self.B, self.N)
        lib().orc_get_state(self._h, C.byref(v))
        st["step_index"] = np.uint64(v.step_index)
        return st

    def set_state(self, st):
        st = {k: (a if k == "step_index" else np.ascontiguousarray(a)) for k, a in st.items()}
        v = make_view(st, self.B, self.N)
        lib().orc_set_state(self._h, C.byref(v))

    def episode_stats(self):
        ret, ln = np.zeros(self.B), np.zeros(self.B, dtype=np.int32)
        lib().orc_episode_stats(self._h, ret.ctypes.data, ln.ctypes.data)
        return ret, ln

    def nearest(self):
        """(distance to the nearest swept intruder of the last step, value-was-float64 flag)."""
        d, f = np.zeros(self.B), np.zeros(self.B, dtype=np.uint8)
        lib().orc_nearest(self._h, d.ctypes.data, f.ctypes.data)
        return d, f

    @property
    def spawn_count(self):
        return lib().orc_spawn_count(self._h)

    @property
    def redraw_count(self):
        return lib().orc_redraw_count(self._h)


def np_random_sample(seed, n):
    out = np.zeros(n)
    lib().orc_np_seed_random_sample(seed, n, out.ctypes.data)
    return out


def np_normal(seed, n, loc=0.0, scale=1.0):
    out = np.zeros(n)
    lib().orc_np_seed_normal(seed, n, loc, scale, out.ctypes.data)
    return out


def np_state(seed, ndraw):
    key = np.zeros(624, dtype=np.uint32)
    pos = C.c_int(0)
    lib().orc_np_seed_state(seed, ndraw, key.ctypes.data, C.byref(pos))
    return key, pos.value


def philox4x32_10(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def pmath_sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_pmath_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def pmath_log(x):
    return lib().orc_pmath_log(float(x))


def mcts_config(N, **over):
    cfg = UamMctsConfig()
    lib().uam_mcts_config_default(C.byref(cfg), N)
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def her_goal_reward(cfg, achieved, desired):
    """compute_reward(achieved_goal, desired_goal) row by row (float64)."""
    a, g = np.asarray(achieved, dtype=np.float64).reshape(-1, 2), np.asarray(desired, dtype=np.float64).reshape(-1, 2)
    return np.asarray([lib().orc_her_goal_reward(C.byref(cfg), a[i, 0], a[i, 1], g[i, 0], g[i, 1]) for i in range(len(a))])


def her_input_reward(cfg, inputs, math_impl=MATH_LIBM):
    """compute_input_reward(new_inputs) row by row (float64)."""
    x = np.ascontiguousarray(inputs, dtype=np.float64)
    x = x.reshape(-1, x.shape[-1])
    return np.asarray([lib().orc_her_input_reward(C.byref(cfg), math_impl, x[i].ctypes.data, x.shape[1]) for i in range(len(x))])


def mcts_move(cfg, state, action, rng_mode=RNG_MT19937, math_impl=MATH_LIBM, np_seed=0, root_id=0, sim=0, move=0):
    """SingleAircraftState(state).move(action) -> (next_state, flags[3], reward, mt_pos)."""
    st = np.ascontiguousarray(state, dtype=np.float64)
    nxt = np.zeros_like(st)
    flags = np.zeros(3, dtype=np.uint8)
    rew, pos = C.c_double(0), C.c_int(0)
    lib().orc_mcts_move(C.byref(cfg), rng_mode, math_impl, np_seed, root_id, sim, move, st.ctypes.data,
                        int(action[0]), int(action[1]), nxt.ctypes.data, flags.ctypes.data, C.byref(rew), C.byref(pos))
    return nxt, flags, rew.value, pos.value


def mcts_best_action(cfg, root_state, rng_mode=RNG_MT19937, math_impl=MATH_LIBM, np_seed=0, root_id=0):
    """MCTS(root).best_action(cfg.simulations, cfg.search_depth) -> dict."""
    st = np.ascontiguousarray(root_state, dtype=np.float64)
    best = np.zeros(2, dtype=np.int32)
    ca = np.zeros((9, 2), dtype=np.int32)
    cq, cn = np.zeros(9), np.zeros(9)
    rn, nn, pos = C.c_double(0), C.c_int(0), C.c_int(0)
    lib().orc_mcts_best_action(C.byref(cfg), rng_mode, math_impl, np_seed, root_id, st.ctypes.data, best.ctypes.data,
                               ca.ctypes.data, cq.ctypes.data, cn.ctypes.data, C.byref(rn), C.byref(nn), C.byref(pos))
    return dict(best_action=best, child_action=ca, child_q=cq, child_n=cn, root_n=rn.value, n_nodes=nn.value,
                rng_pos=pos.value)
