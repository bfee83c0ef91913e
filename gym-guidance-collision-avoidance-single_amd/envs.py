"""Host-side mirror of the reference's env interface over the HIP C-ABI.

Two shapes, same kernels:

* `BatchedSingleAircraftEnv` — VecEnv-shaped (baselines/common/vec_env/__init__.py:26-105):
  `num_envs`, `reset() -> obs[B, D]`, `step_async/step_wait/step -> (obs, rews, dones, infos)`,
  auto-reset on done (dummy_vec_env.py:52-55).  Observations, rewards and dones are torch
  tensors that stay on the GPU; `infos` is materialised lazily.
* `SingleAircraftEnv` & variants — gym.Env duck types with the reference's signatures
  (Simulators/SingleAircraftEnv.py:27-139): `reset() -> obs`, `step(a) -> (obs, reward, done,
  {'result': code})`, attributes `observation_space`, `action_space`, `no_conflict`,
  `intruder_size`.  They default to the NumPy-compatible MT19937 stream and exact float64
  promotion, so `env.seed(s)` reproduces the reference run under `np.random.seed(s)`.

PyTorch is used only for device buffers and streams.  There is no CPU path: constructing an
env without the built library or without a GPU raises.
"""
import ctypes as C
import math

import numpy as np

from . import _capi as capi
from . import spaces


class Config:
    """Editable mirror of Simulators/config.py:4-55; snapshotted into `uam_config` when an env is built."""
    window_width = 800
    window_height = 800
    diagonal = 800
    intruder_size = 80
    EPISODES = 1000
    G = 9.8
    tick = 30
    scale = 30
    minimum_separation = 555 / scale
    NMAC_dist = 150 / scale
    horizon_dist = 4000 / scale
    initial_min_dist = 3000 / scale
    goal_radius = 600 / scale
    min_speed = 50 / scale
    max_speed = 80 / scale
    d_speed = 5 / scale
    speed_sigma = 2 / scale
    position_sigma = 10 / scale
    d_heading = math.radians(5)
    heading_sigma = math.radians(4)
    max_steps = 1000
    NMAC_penalty = -10 / 10
    conflict_penalty = -5 / 10
    wall_penalty = -5 / 10
    step_penalty = -0.01 / 10
    goal_reward = 10 / 10
    sparse_reward = False
    conflict_coeff = 0.00025
    n = 4


class PackagedConfig(Config):
    """The constants the packaged (deprecated) copy runs with: gym_guidance_collision_avoidance_single/envs/config.py
    (heading_sigma 2 degrees, intruder_size 0: a map with the ownship and a goal only, which is what
    `PackagedSingleAircraft2Env()` then runs — pass num_intruders for more) and the literals of its
    SingleAircraft2Env._terminal_reward (:155-176): NMAC -5, conflict -1, out of map -100 and terminal, goal +1."""
    intruder_size = 0
    heading_sigma = math.radians(2)
    NMAC_penalty = -5
    conflict_penalty = -1
    wall_penalty = -100
    goal_reward = 1
    wall_terminates = True


_VARIANTS = {
    # name: (obs_mode, action_mode, dynamics, random_own_start)
    "discrete9": (capi.OBS_NORM4, capi.ACT_DISCRETE9, capi.DYN_STRAIGHT, 0),    # Simulators/SingleAircraftEnv.py
    "continuous": (capi.OBS_NORM4, capi.ACT_CONTINUOUS1, capi.DYN_STRAIGHT, 0),  # Simulators/SingleAircraft2Env.py
    # the packaged copy's action semantics (un-offset heading command, live speed command); its reward constants and
    # out-of-map rule come with `config=PackagedConfig` (gym_guidance_collision_avoidance_single/envs/SingleAircraft2Env.py)
    "continuous2": (capi.OBS_NORM4, capi.ACT_CONTINUOUS2, capi.DYN_STRAIGHT, 0),
    "mcts": (capi.OBS_RAW4, capi.ACT_PAIR, capi.DYN_STRAIGHT, 0),                # Simulators/SingleAircraftMCTSEnv.py
    "randintruder": (capi.OBS_RAW6, capi.ACT_PAIR, capi.DYN_RANDINTRUDER, 0),    # ...MCTSRandIntruderEnv.py
    "random_start": (capi.OBS_NORM4, capi.ACT_DISCRETE9, capi.DYN_STRAIGHT, 1),  # Simulators/SingleAircraftRandomEnv.py
    "her9": (capi.OBS_KNN5, capi.ACT_DISCRETE9, capi.DYN_STRAIGHT, 1),           # ...Discrete9HEREnv.py (k-nearest obs)
    "her3": (capi.OBS_KNN5, capi.ACT_DISCRETE3, capi.DYN_STRAIGHT, 1),           # ...Discrete3HEREnv.py (+ shaped reward)
}
_RNG = {"philox": capi.RNG_PHILOX, "mt19937": capi.RNG_MT19937, "mt19937_shared": capi.RNG_MT19937_SHARED}
_PREC = {"f32": capi.POS_F32, "exact": capi.POS_EXACT}


_FUSE = {"auto": 0, "always": 1, "never": 2}


def make_config(num_envs, num_intruders=None, variant="discrete9", rng="philox", precision="f32", seed=0,
                auto_reset=True, env_id_offset=0, config=Config, **overrides):
    """Build a `uam_config` from the (possibly edited) `Config` class plus explicit field overrides."""
    obs_mode, action_mode, dynamics, ros = _VARIANTS[variant]
    n = int(config.intruder_size if num_intruders is None else num_intruders)
    cfg = capi.default_config(
        int(num_envs), n, obs_mode=obs_mode, action_mode=action_mode, dynamics=dynamics, random_own_start=ros,
        rng_mode=_RNG[rng], precision=_PREC[precision], auto_reset=int(bool(auto_reset)),
        seed=int(seed) & 0xFFFFFFFFFFFFFFFF, env_id_offset=int(env_id_offset),
        sparse_reward=int(bool(config.sparse_reward)),
        window_width=config.window_width, window_height=config.window_height,
        minimum_separation=config.minimum_separation, nmac_dist=config.NMAC_dist,
        initial_min_dist=config.initial_min_dist, goal_radius=config.goal_radius,
        min_speed=config.min_speed, max_speed=config.max_speed, d_speed=config.d_speed,
        speed_sigma=config.speed_sigma, position_sigma=config.position_sigma,
        d_heading=config.d_heading, heading_sigma=config.heading_sigma,
        nmac_penalty=config.NMAC_penalty, conflict_penalty=config.conflict_penalty,
        step_penalty=config.step_penalty, goal_reward=config.goal_reward,
        own_speed0=config.min_speed, wall_penalty=config.wall_penalty, diagonal=config.diagonal, knn=int(config.n),
        # the HER env ends the episode when the ownship leaves the map, but only under sparse_reward (:217-221)
        wall_terminates=int((variant in ("her9", "her3") and bool(config.sparse_reward)) or bool(getattr(config, "wall_terminates", False))),
        conflict_coeff=config.conflict_coeff,
        # Discrete3HEREnv.py: goal 100 px inside the map (:349-353), reward shaped by the nearest intruder (:225-232)
        goal_margin=100.0 if variant == "her3" else 0.0, shaped_reward=int(variant == "her3"))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError("uam_config has no field %r" % k)
        if k == "fuse_rollout" and isinstance(v, str):
            v = _FUSE[v]
        setattr(cfg, k, v)
    return cfg


class LazyInfos:
    """Sequence of `{'result': code}` dicts (SingleAircraftEnv.py:139) backed by the device code tensor."""

    def __init__(self, codes, as_dict=True, episodes=None):
        self.codes = codes
        self._as_dict = as_dict
        self._host = None
        self._episodes = episodes     # callable -> (done[B], return[B], length[B]) host arrays, or None

    def _h(self):
        if self._host is None:
            self._host = self.codes.cpu().numpy()
        return self._host

    def __len__(self):
        return int(self.codes.shape[0])

    def __getitem__(self, i):
        code = capi.INFO_CODES[int(self._h()[i])]
        if not self._as_dict:
            return code
        info = {"result": code}
        if self._episodes is not None:
            if callable(self._episodes):
                self._episodes = self._episodes()
            done, ret, ln = self._episodes
            if done[i]:   # Monitor.step: info['episode'] = {'r','l','t'}, baselines/bench/monitor.py:68-74
                info["episode"] = {"r": round(float(ret[i]), 6), "l": int(ln[i])}
        return info

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def copy(self):
        return list(self)


class BatchedSingleAircraftEnv:
    """B envs stepped by one fused HIP kernel per `step`; VecEnv-shaped.

    Buffer ownership (differs from the reference, which returns a fresh array per call): `reset`, `step`, `step_wait`,
    `observe` and `step_many` return the env's OWN device tensors (`obs`, `rews`, `dones`) and a lazy view of its info
    codes.  They are valid until the next call that steps or resets the env, which overwrites them in place — a
    rollout that keeps step t's tensors must `.clone()` them, or use `rollout_into` / `bind_outputs` to have each
    step written straight into its slot of a caller-owned slab.  `LazyInfos` reads the code buffer on first access:
    materialise it (`list(infos)` / `infos.copy()`) before stepping again if it is needed later.  `UamVecEnv` (NumPy
    facing) alternates two host buffers and has none of this."""

    def __init__(self, num_envs, num_intruders=None, variant="discrete9", rng="philox", precision="f32", seed=0,
                 auto_reset=True, device=None, env_id_offset=0, config=Config, cfg=None, monitor=False, **overrides):
        import torch
        self._torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("uamenv needs a HIP GPU (torch.cuda.is_available() is False); there is no CPU backend")
        self._lib = capi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        self.cfg = cfg if cfg is not None else make_config(
            num_envs, num_intruders, variant, rng, precision, seed, auto_reset, env_id_offset, config,
            track_episodes=int(bool(monitor)), **overrides)
        self.monitor = bool(self.cfg.track_episodes)
        self.num_envs = int(self.cfg.num_envs)
        self.intruder_size = int(self.cfg.num_intruders)
        self.obs_dim = capi.obs_dim(self.cfg)
        # observation_space advertises 4N+8 even for the 6-wide RandIntruder obs (RandIntruderEnv.py:51-52);
        # the HER env advertises len(observation) + 2 (Discrete9HEREnv.py:42)
        nobs = (4 + 5 * int(self.cfg.knn) + 2) if self.cfg.obs_mode == capi.OBS_KNN5 else self.intruder_size * 4 + 8
        self.observation_space = spaces.Box(low=-1000, high=1000, shape=(nobs,), dtype=np.float32)
        if self.cfg.action_mode == capi.ACT_CONTINUOUS1:
            self.action_space = spaces.Box(low=-1, high=1, shape=(1,), dtype=np.float64)
        elif self.cfg.action_mode == capi.ACT_CONTINUOUS2:
            self.action_space = spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float64)
        else:
            self.action_space = spaces.Discrete(3 if self.cfg.action_mode == capi.ACT_DISCRETE3 else 9)
        self._h = C.c_void_p()
        capi.check(self._lib.uam_create(C.byref(self.cfg), self.device.index, C.byref(self._h)))
        B, D = self.num_envs, self.obs_dim
        with torch.cuda.device(self.device):
            self._obs = torch.empty((B, D), dtype=torch.float32, device=self.device)
            self._rew = torch.empty((B,), dtype=torch.float32, device=self.device)
            self._done = torch.empty((B,), dtype=torch.uint8, device=self.device)
            self._info = torch.empty((B,), dtype=torch.uint8, device=self.device)
        self._pending = None
        self._act_numel = self.num_envs * (2 if self.cfg.action_mode in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else 1)
        # fixed argument objects of the per-step call, built once
        self._step_fn = self._lib.uam_step
        self._fixed_ptrs = tuple(C.c_void_p(t.data_ptr()) for t in (self._obs, self._rew, self._done, self._info))
        self.closed = False

    def bind_outputs(self, rew, done, info, obs=None):
        """Make `step` write rewards / dones / info codes (and optionally the observations) into caller-owned device
        tensors (float32[B], uint8[B], uint8[B], float32[B, D]) — e.g. slices of one slab that leaves the device in
        a single copy."""
        torch = self._torch
        for t, dt in ((rew, torch.float32), (done, torch.uint8), (info, torch.uint8)):
            if t.dtype != dt or t.numel() != self.num_envs or not t.is_contiguous() or t.device != self.device:
                raise ValueError("bind_outputs needs contiguous device tensors float32[B], uint8[B], uint8[B]")
        if obs is not None:
            if obs.dtype != torch.float32 or tuple(obs.shape) != (self.num_envs, self.obs_dim) or not obs.is_contiguous() \
                    or obs.device != self.device:
                raise ValueError("bind_outputs: obs must be a contiguous device tensor float32[B, D]")
            self._obs = obs
        self._rew, self._done, self._info = rew, done, info
        self._fixed_ptrs = tuple(C.c_void_p(t.data_ptr()) for t in (self._obs, self._rew, self._done, self._info))

    # ---- plumbing
    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _as_actions(self, actions):
        torch = self._torch
        am = self.cfg.action_mode
        dt = torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32
        if isinstance(actions, torch.Tensor) and actions.dtype == dt and actions.device == self.device \
                and actions.is_contiguous() and actions.numel() == self._act_numel:
            return actions          # fast path: a ready device tensor (the per-step host cost matters for small batches)
        if not isinstance(actions, torch.Tensor):
            actions = torch.as_tensor(np.asarray(actions), dtype=dt)
        a = actions.to(device=self.device, dtype=dt).contiguous()
        want = self.num_envs * (2 if am in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else 1)
        if a.numel() != want:
            raise ValueError("actions has %d elements, expected %d for %d envs" % (a.numel(), want, self.num_envs))
        return a

    def close(self):
        if not self.closed and self._h:
            self._lib.uam_destroy(self._h)
            self._h = C.c_void_p()
        self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- VecEnv API
    def seed(self, seeds=None):
        """Re-seed.  MT19937 handles: one seed per env (scalar = the same for all; array [B]), each in [0, 2**32) like
        `np.random.seed` (which raises beyond that); None = cfg.seed + global env index.  Philox handles have ONE key
        for the whole batch (the env index is part of the counter), so only a scalar seed makes sense there."""
        if seeds is None:
            capi.check(self._lib.uam_seed(self._h, None))
            return seeds
        raw = np.asarray(seeds)
        if raw.size == 0 or (raw.dtype.kind not in "ui") or (raw.dtype.kind == "i" and (raw < 0).any()):
            raise ValueError("seeds must be non-negative integers")
        if self.cfg.rng_mode in (capi.RNG_MT19937, capi.RNG_MT19937_SHARED):
            if (raw.astype(np.uint64) >= np.uint64(1) << np.uint64(32)).any():
                raise ValueError("Seed must be between 0 and 2**32 - 1")        # np.random.seed's own message
        elif raw.size > 1 and len(np.unique(raw)) > 1:
            raise ValueError("a Philox handle has one key for the whole batch: pass a scalar seed "
                             "(per-env streams come from the env index in the counter)")
        arr = np.ascontiguousarray(np.broadcast_to(raw.astype(np.uint64), (self.num_envs,)))
        capi.check(self._lib.uam_seed(self._h, arr.ctypes.data))
        return seeds

    def reset(self, mask=None):
        torch = self._torch
        m = None
        if mask is not None:
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
        capi.check(self._lib.uam_reset(self._h, None if m is None else m.data_ptr(), self._obs.data_ptr(), self._stream()))
        return self._obs

    def step_async(self, actions):
        if self._pending is not None:
            raise RuntimeError("already running an async step")
        a = self._as_actions(actions)
        o, r, d, i = self._fixed_ptrs
        rc = self._step_fn(self._h, a.data_ptr(), o, r, d, i, self._stream())
        if rc:
            capi.check(rc)
        self._pending = a   # keep the action buffer alive until the kernel has consumed it

    def step_into(self, actions, rew_out, done_out):
        """`step` that writes rewards (float32[B]) and dones (uint8[B]) straight into caller-owned device
        rows, e.g. row k of a [K, B] rollout slab that is gathered across GPUs once per chunk."""
        torch = self._torch
        a = self._as_actions(actions)
        if rew_out.dtype != torch.float32 or done_out.dtype != torch.uint8 or rew_out.numel() != self.num_envs \
                or done_out.numel() != self.num_envs or not (rew_out.is_contiguous() and done_out.is_contiguous()):
            raise ValueError("rew_out must be contiguous float32[B] and done_out contiguous uint8[B]")
        capi.check(self._lib.uam_step(self._h, a.data_ptr(), self._obs.data_ptr(), rew_out.data_ptr(),
                                      done_out.data_ptr(), self._info.data_ptr(), self._stream()))
        self._keep = a
        return self._obs

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError("not running an async step")
        self._pending = None
        eps = None
        if self.monitor:
            def eps():
                ret, ln = self.episode_stats()
                return self._done.cpu().numpy(), ret.cpu().numpy(), ln.cpu().numpy()
        return self._obs, self._rew, self._done.view(self._torch.bool), LazyInfos(self._info, episodes=eps)   # 0/1 bytes: zero-copy view

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_many(self, action_tape, want_obs=True):
        """K steps from an action tape [K, B(,2)] with no host round trips; returns
        (obs after the last step, rewards [K,B], dones [K,B] uint8, info codes [K,B] uint8)."""
        torch = self._torch
        am = self.cfg.action_mode
        dt = torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32
        tape = torch.as_tensor(action_tape).to(device=self.device, dtype=dt).contiguous()
        K = int(tape.shape[0])
        if tape.numel() != K * self.num_envs * (2 if am in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else 1):
            raise ValueError("action tape has the wrong shape %s" % (tuple(tape.shape),))
        rew = torch.empty((K, self.num_envs), dtype=torch.float32, device=self.device)
        done = torch.empty((K, self.num_envs), dtype=torch.uint8, device=self.device)
        info = torch.empty((K, self.num_envs), dtype=torch.uint8, device=self.device)
        capi.check(self._lib.uam_step_many(self._h, K, tape.data_ptr(), self._obs.data_ptr() if want_obs else None,
                                           rew.data_ptr(), done.data_ptr(), info.data_ptr(), self._stream()))
        return self._obs, rew, done, info

    def _check_rollout_buffers(self, K, action_tape, rew_out, done_out, info_out, obs_out, actions_are_output=False):
        """Every buffer a chunk call hands to the C side, checked here: the library cannot know their sizes, so an
        undersized, strided, wrongly typed or host tensor would be an out-of-bounds device write.  Returns obs_out
        (the env's own [1, B, D] view when None).  action_tape may be None only when it is an output."""
        torch = self._torch
        am = self.cfg.action_mode
        dt = torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32
        if K < 1:
            raise ValueError("K must be >= 1")
        if action_tape is None and not actions_are_output:
            raise ValueError("an action tape is required")
        if action_tape is not None and (action_tape.dtype != dt or action_tape.device != self.device or not action_tape.is_contiguous()
                                        or action_tape.numel() != K * self._act_numel):
            raise ValueError("%s must be a contiguous device tensor [K, B(,2)] of %s" % ("actions_out" if actions_are_output else "action tape", dt))
        if rew_out is None or done_out is None:
            raise ValueError("rew_out and done_out are required")
        for t, tdt in ((rew_out, torch.float32), (done_out, torch.uint8), (info_out, torch.uint8)):
            if t is not None and (t.dtype != tdt or t.numel() != K * self.num_envs or not t.is_contiguous()
                                  or t.device != self.device):
                raise ValueError("rollout slabs must be contiguous device tensors [K, B]")
        if obs_out is None:
            obs_out = self._obs.view(1, self.num_envs, self.obs_dim)
        if obs_out.dtype != torch.float32 or obs_out.dim() != 3 or tuple(obs_out.shape[1:]) != (self.num_envs, self.obs_dim) \
                or not obs_out.is_contiguous() or obs_out.device != self.device or int(obs_out.shape[0]) < 1:
            raise ValueError("obs_out must be a contiguous device tensor float32[R, B, D]")
        return obs_out

    def rollout_into(self, action_tape, rew_out, done_out, info_out=None, obs_out=None):
        """K steps from a device action tape [K, B(,2)] in ONE C call (uam_rollout) that keeps every step's
        observation: rewards / dones / info codes go to caller-owned device slabs [K, B]; step k's observation goes
        to slot k % R of `obs_out` (float32[R, B, D]; default: the env's own [B, D] buffer, R = 1).  The open-loop
        rollout of baselines/ppo2/runner.py:26-38 with the actions already on the device.  Returns obs_out."""
        K = int(action_tape.shape[0])
        obs_out = self._check_rollout_buffers(K, action_tape, rew_out, done_out, info_out, obs_out)
        capi.check(self._lib.uam_rollout(self._h, K, action_tape.data_ptr(), obs_out.data_ptr(), int(obs_out.shape[0]),
                                         rew_out.data_ptr(), done_out.data_ptr(),
                                         None if info_out is None else info_out.data_ptr(), self._stream()))
        self._keep = action_tape
        return obs_out

    def rollout_policy(self, K, kind="goal", seed=1, rew_out=None, done_out=None, info_out=None, obs_out=None, actions_out=None):
        """K closed-loop steps with a built-in device-side policy ('random' | 'goal') choosing a fresh action before
        every step (uam_rollout_policy).  Returns (obs_out [R,B,D], rewards [K,B], dones [K,B] uint8, infos [K,B],
        actions [K,B(,2)])."""
        torch = self._torch
        K = int(K)
        B, dev = self.num_envs, self.device
        am = self.cfg.action_mode
        if rew_out is None:
            rew_out = torch.empty((K, B), dtype=torch.float32, device=dev)
        if done_out is None:
            done_out = torch.empty((K, B), dtype=torch.uint8, device=dev)
        if info_out is None:
            info_out = torch.empty((K, B), dtype=torch.uint8, device=dev)
        if actions_out is None:
            actions_out = torch.empty((K, B, 2) if am in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else (K, B),
                                      dtype=torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32, device=dev)
        obs_out = self._check_rollout_buffers(K, actions_out, rew_out, done_out, info_out, obs_out, actions_are_output=True)
        capi.check(self._lib.uam_rollout_policy(self._h, K, {"random": 0, "goal": 1}[kind], int(seed), obs_out.data_ptr(),
                                                int(obs_out.shape[0]), rew_out.data_ptr(), done_out.data_ptr(),
                                                info_out.data_ptr(), actions_out.data_ptr(), self._stream()))
        return obs_out, rew_out, done_out, info_out, actions_out

    def steps_per_launch(self, K):
        """K if a tape-driven chunk of K steps (step_many / rollout_into) runs as one launch on this handle, else 1."""
        return int(self._lib.uam_steps_per_launch(self._h, int(K)))

    def rollout_kernel(self, K):
        """Name of the kernel a tape-driven chunk of K steps runs in on this handle (labels of bench lines / profiles)."""
        return self._lib.uam_rollout_kernel(self._h, int(K)).decode()

    def policy_steps_per_launch(self, K):
        """The same for rollout_policy: K only where the policy is evaluated inside the looped kernel's launch."""
        return int(self._lib.uam_policy_steps_per_launch(self._h, int(K)))

    def bind_rollout(self, action_tape, rew_out, done_out, info_out=None, obs_out=None):
        """`rollout_into` with the argument checks and pointer extraction done ONCE: returns a zero-argument callable
        that only makes the C call on the stream that is current now.  For hot loops that replay the same buffers
        (a rollout slab ring): the checks cost ~25 us of Python per call, a tenth of a 20-step chunk at 65536 envs."""
        K = int(action_tape.shape[0])
        obs_out = self._check_rollout_buffers(K, action_tape, rew_out, done_out, info_out, obs_out)
        fn, h = self._lib.uam_rollout, self._h
        args = (h, K, C.c_void_p(action_tape.data_ptr()), C.c_void_p(obs_out.data_ptr()), int(obs_out.shape[0]),
                C.c_void_p(rew_out.data_ptr()), C.c_void_p(done_out.data_ptr()),
                C.c_void_p(info_out.data_ptr()) if info_out is not None else None, self._stream())
        keep = (action_tape, rew_out, done_out, info_out, obs_out)

        def call(_keep=keep):
            rc = fn(*args)
            if rc:
                capi.check(rc)
        return call

    def observe(self):
        capi.check(self._lib.uam_observe(self._h, self._obs.data_ptr(), self._stream()))
        return self._obs

    def policy_actions(self, kind="random", seed=1, out=None):
        """Device-side action generator for benchmarks: 'random' or 'goal' (turn to goal, accelerate)."""
        torch = self._torch
        am = self.cfg.action_mode
        if out is None:
            shape = (self.num_envs, 2) if am in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else (self.num_envs,)
            out = torch.empty(shape, dtype=torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32, device=self.device)
        capi.check(self._lib.uam_policy(self._h, {"random": 0, "goal": 1}[kind], int(seed), out.data_ptr(), self._stream()))
        return out

    def episode_stats(self, ret=None, ln=None):
        """(return float64[B], length int32[B]) of each env's most recently finished episode (monitor=True)."""
        torch = self._torch
        if ret is None:
            ret = torch.empty((self.num_envs,), dtype=torch.float64, device=self.device)
        if ln is None:
            ln = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        capi.check(self._lib.uam_episode_stats(self._h, ret.data_ptr(), ln.data_ptr(), self._stream()))
        return ret, ln

    def nearest(self):
        """float32[B]: distance to the nearest intruder swept by the last step (variant "her3": what
        Discrete3HEREnv.step returns as its 4th value)."""
        out = self._torch.empty((self.num_envs,), dtype=self._torch.float32, device=self.device)
        capi.check(self._lib.uam_nearest(self._h, out.data_ptr(), self._stream()))
        return out

    @property
    def no_conflict(self):
        out = self._torch.empty((self.num_envs,), dtype=self._torch.int32, device=self.device)
        capi.check(self._lib.uam_counters(self._h, out.data_ptr(), self._stream()))
        return out

    @property
    def step_bytes(self):
        return int(self._lib.uam_step_bytes(self._h))

    def rollout_bytes(self, K):
        """Algorithmic HBM bytes of one K-step rollout call as this handle runs it (looped kernel: state once)."""
        return int(self._lib.uam_rollout_bytes(self._h, int(K)))

    def timing_begin(self):
        capi.check(self._lib.uam_timing_begin(self._h, self._stream()))

    def timing_end(self):
        ms = C.c_float(0)
        capi.check(self._lib.uam_timing_end(self._h, self._stream(), C.byref(ms)))
        return float(ms.value)

    def kernel_timing(self, enable=True):
        """Bracket each launch of the following steps with hipEvents (diagnostic; serialises launches)."""
        capi.check(self._lib.uam_kernel_timing(self._h, int(bool(enable))))

    def kernel_timing_read(self):
        """-> (ownship-kernel ms, sweep-kernel ms, sampled steps) averaged over the sampled steps."""
        a, b, n = C.c_float(0), C.c_float(0), C.c_int32(0)
        capi.check(self._lib.uam_kernel_timing_read(self._h, self._stream(), C.byref(a), C.byref(b), C.byref(n)))
        return float(a.value), float(b.value), int(n.value)

    # ---- state exchange (checkpoint / parity injection)
    def get_state(self, fields=None):
        B, N = self.num_envs, self.intruder_size
        names = list(capi.STATE_FIELDS) if fields is None else list(fields)
        if self.cfg.rng_mode not in (capi.RNG_MT19937, capi.RNG_MT19937_SHARED):
            names = [n for n in names if not n.startswith("mt_")]
        hold_sh = self.cfg.dynamics == capi.DYN_RANDINTRUDER or self.cfg.obs_mode == capi.OBS_RAW6
        if not hold_sh:
            names = [n for n in names if n not in ("int_speed", "int_heading")]
        st = {}
        for n in names:
            dt, shape = capi.STATE_FIELDS[n]
            st[n] = np.zeros({"B": (B,), "BN": (B, N), "B624": (B, 624)}[shape], dtype=dt)
        v = capi.make_view(st, B, N)
        capi.check(self._lib.uam_get_state(self._h, C.byref(v)))
        st["step_index"] = np.uint64(v.step_index)
        return st

    def set_state(self, st):
        B, N = self.num_envs, self.intruder_size
        arrs = {}
        for n, a in st.items():
            if n == "step_index":
                arrs[n] = int(a)
                continue
            dt, _ = capi.STATE_FIELDS[n]
            arrs[n] = np.ascontiguousarray(a, dtype=dt)
        v = capi.make_view(arrs, B, N)
        capi.check(self._lib.uam_set_state(self._h, C.byref(v)))


class GraphedLoop:
    """A closed acting loop captured ONCE into a HIP graph and replayed: `policy(obs) -> actions` followed by
    `env.step`, `steps` times per replay, with no host round trip between policy and env or between steps — the loop
    of baselines/ppo2/runner.py:26-38 / Algorithms/MCTS/Agent_RandInt.py:36-46 for batches small enough to be
    launch-bound.  Needs an env built with `graph_safe=True` (the step counter of the random draws then lives in
    device memory, so the captured kernels' arguments stay valid for every replay).

    `policy` must be capturable: device-side torch ops on the static observation buffer, or `env.policy_actions(kind,
    out=...)`; it returns a device tensor in the env's action layout.  After `replay()` the results of the last
    `steps` steps are in `rewards[steps, B]`, `dones[steps, B]` (uint8), `infos[steps, B]` (codes) and `obs` (after
    the last step) — static buffers that the next replay overwrites."""

    def __init__(self, env, policy, steps=1):
        torch = env._torch
        if not int(env.cfg.graph_safe):
            raise ValueError("GraphedLoop needs an env created with graph_safe=True")
        self.env, self.steps = env, int(steps)
        B, dev = env.num_envs, env.device
        self.rewards = torch.zeros((self.steps, B), dtype=torch.float32, device=dev)
        self.dones = torch.zeros((self.steps, B), dtype=torch.uint8, device=dev)
        self.infos = torch.zeros((self.steps, B), dtype=torch.uint8, device=dev)
        self.obs = env._obs
        self._policy = policy
        # torch's capture protocol: one eager run on a side stream first (lazy initialisations happen there), with the
        # env state put back afterwards so that the capture itself does not advance the episode
        st = env.get_state()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        env.set_state(st)
        env.observe()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        torch.cuda.synchronize(dev)
        env.set_state(st)           # a capture only records: nothing ran, but keep the contract obvious
        env.observe()

    def _body(self):
        env = self.env
        for j in range(self.steps):
            a = self._policy(self.obs)
            capi.check(env._lib.uam_step(env._h, a.data_ptr(), self.obs.data_ptr(), self.rewards[j].data_ptr(),
                                         self.dones[j].data_ptr(), self.infos[j].data_ptr(), env._stream()))
            self._keep = a

    def replay(self):
        self.graph.replay()
        return self.obs, self.rewards, self.dones, self.infos


class SingleAircraftEnv:
    """Drop-in for Simulators/SingleAircraftEnv.py (one env, NumPy in/out, gym signatures)."""
    _variant = "discrete9"
    _info_as_dict = True
    metadata = {"render.modes": ["human"]}

    def __init__(self, num_intruders=None, seed=None, device=None, rng="mt19937", precision="exact", config=Config,
                 **overrides):
        self._seed = 2 if seed is None else int(seed)   # the reference calls self.seed(2) in __init__ (:41)
        self._vec = BatchedSingleAircraftEnv(1, num_intruders, self._variant, rng, precision, self._seed,
                                             auto_reset=False, device=device, config=config, **overrides)
        self.intruder_size = self._vec.intruder_size
        self.observation_space = self._vec.observation_space
        self.action_space = self._vec.action_space
        self.window_width, self.window_height = config.window_width, config.window_height
        self.viewer = None
        self.state = None
        # one step = one launch sequence + ONE host synchronisation: reward, done, info and no_conflict live in a
        # 16-byte device slab next to the observation row, and both leave through page-locked buffers
        torch = self._vec._torch
        v = self._vec
        nb = 4 * v.obs_dim
        self._d_slab = torch.zeros((nb + 16,), dtype=torch.uint8, device=v.device)      # obs row | reward | done | info
        self._h_slab = torch.zeros((nb + 16,), dtype=torch.uint8, pin_memory=True)
        self._h_obs = self._h_slab[:nb].view(torch.float32).view(1, v.obs_dim)
        v.bind_outputs(self._d_slab[nb:nb + 4].view(torch.float32), self._d_slab[nb + 4:nb + 5], self._d_slab[nb + 5:nb + 6],
                       obs=self._d_slab[:nb].view(torch.float32).view(1, v.obs_dim))
        self._nb = nb
        # the action goes up through a page-locked word as well (a pageable torch.as_tensor(...).to(device) is a
        # synchronous copy of its own)
        am = v.cfg.action_mode
        act_dt = torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32
        act_shape = (1, 2) if am in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else (1,)
        self._h_act = torch.zeros(act_shape, dtype=act_dt, pin_memory=True)
        self._d_act = torch.zeros(act_shape, dtype=act_dt, device=v.device)
        self._h_side = torch.zeros((2,), dtype=torch.int32, pin_memory=True)   # no_conflict | nearest distance bits
        self._no_conflict = 0

    def _fetch(self, obs):
        """Bring the observation row and the scalar slab to the host with one synchronisation."""
        v = self._vec
        capi.check(v._lib.uam_counters(v._h, self._h_side.data_ptr(), v._stream()))     # straight into host memory
        self._h_slab.copy_(self._d_slab, non_blocking=True)          # observation row and scalars in one copy
        v._torch.cuda.current_stream(v.device).synchronize()
        h = self._h_slab.numpy()[self._nb:]
        self._no_conflict = int(self._h_side[0])
        return float(h[0:4].view(np.float32)[0]), bool(h[4]), int(h[5])

    def seed(self, seed=None):
        """Unlike the reference (whose per-env np_random is never used, :43-45) this really seeds the env:
        seed(s) ≙ np.random.seed(s) in the reference's process."""
        self._seed = int(seed) if seed is not None else 0
        self._vec.seed([self._seed])
        return [seed]

    def _obs_np(self, t):
        return t[0].numpy().astype(np.float64)

    def reset(self):
        self._fetch(self._vec.reset())
        return self._obs_np(self._h_obs)

    def _encode_action(self, action):
        return [int(action)]

    def _step_host(self, action):
        self._h_act.numpy()[...] = np.asarray(self._encode_action(action)).reshape(self._h_act.shape)
        self._d_act.copy_(self._h_act, non_blocking=True)
        obs, _, _, _ = self._vec.step(self._d_act)
        rew, done, code = self._fetch(obs)
        return self._obs_np(self._h_obs), rew, done, code

    def step(self, action):
        ob, rew, done, code = self._step_host(action)
        info = capi.INFO_CODES[code]
        return ob, rew, done, ({"result": info} if self._info_as_dict else info)

    @property
    def no_conflict(self):
        """Agent.py:61 reads env.no_conflict after the episode: the counter as of the last reset()/step()."""
        return self._no_conflict

    def render(self, mode="human"):
        return None   # no display on the GPU box; SURVEY §2 #8 (out of scope)

    def close(self):
        self._vec.close()


class SingleAircraft2Env(SingleAircraftEnv):
    """Simulators/SingleAircraft2Env.py: continuous action, heading += d_heading*(a-1), no speed command."""
    _variant = "continuous"

    def _encode_action(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        assert a.shape == (1,) and -1 <= a[0] <= 1, "given action is in incorrect shape"   # :127
        return np.asarray([a[0]], dtype=np.float32)


class PackagedSingleAircraft2Env(SingleAircraftEnv):
    """gym_guidance_collision_avoidance_single/envs/SingleAircraft2Env.py (the packaged copy): action (a[0], a[1]) in
    [-1, 1]^2 with heading += d_heading*a[0], speed += d_speed*a[1] (:281-285), its own reward literals and the
    out-of-map termination 'w' (:155-176), info returned as a bare string (:134)."""
    _variant = "continuous2"
    _info_as_dict = False

    def __init__(self, num_intruders=None, seed=None, device=None, rng="mt19937", precision="exact", config=PackagedConfig,
                 **overrides):
        super().__init__(num_intruders, seed, device, rng, precision, config, **overrides)

    def _encode_action(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        assert a.shape == (2,) and (a >= -1).all() and (a <= 1).all(), "given action is in incorrect shape"   # :127
        return np.asarray([[a[0], a[1]]], dtype=np.float32)


class SingleAircraftMCTSEnv(SingleAircraftEnv):
    """Simulators/SingleAircraftMCTSEnv.py: raw observation, (a0, a1) pair action."""
    _variant = "mcts"

    def _encode_action(self, action):
        return np.asarray([[int(action[0]), int(action[1])]], dtype=np.int32)


class SingleAircraftMCTSRandIntruderEnv(SingleAircraftMCTSEnv):
    """Simulators/SingleAircraftMCTSRandIntruderEnv.py: drifting intruders with random heading changes,
    raw 6-wide observation, info returned as a bare string (:164)."""
    _variant = "randintruder"
    _info_as_dict = False


class SingleAircraftRandomEnv(SingleAircraftEnv):
    """Simulators/SingleAircraftRandomEnv.py: ownship start drawn like an intruder."""
    _variant = "random_start"


class SingleAircraftDiscrete9HEREnv(SingleAircraftEnv):
    """Simulators/SingleAircraftDiscrete9HEREnv.py: gym.GoalEnv-style dict observation with the `Config.n`
    nearest intruders (ascending distance), random ownship start, out-of-map termination under
    sparse_reward.  Like the reference, the constructor already runs one reset() (:32)."""
    _variant = "her9"

    def __init__(self, num_intruders=None, seed=None, device=None, rng="mt19937", precision="exact", config=Config,
                 **overrides):
        super().__init__(num_intruders, seed, device, rng, precision, config, **overrides)
        self._config = config
        self.goal_radius = config.goal_radius
        self.minimum_separation = config.minimum_separation
        self.NMAC_dist = config.NMAC_dist
        self.reset()

    def seed(self, seed=None):
        out = super().seed(seed)
        self.reset()          # np.random.seed(s); env = Env() in the reference: the constructor's reset comes first
        return out

    def _obs_np(self, t):
        row = t[0].numpy().astype(np.float64)
        k = 4 + 5 * int(self._vec.cfg.knn)
        return {"observation": row[:k], "achieved_goal": row[k:k + 2], "desired_goal": row[k + 2:k + 4]}

    def compute_reward(self, achieved_goal, desired_goal, info):
        """Discrete9HEREnv.py:229-243 (without its in-place un-normalisation of the arguments)."""
        c = self._config
        a = np.asarray(achieved_goal, dtype=np.float64) * np.array([c.window_width, c.window_height])
        g = np.asarray(desired_goal, dtype=np.float64) * np.array([c.window_width, c.window_height])
        d = np.linalg.norm(a - g, axis=-1)
        if d < self.goal_radius:
            return c.goal_reward
        return c.step_penalty if c.sparse_reward else -d / 1200

    def compute_input_reward(self, new_inputs):
        """Discrete9HEREnv.py:245-276."""
        c = self._config
        x = np.asarray(new_inputs, dtype=np.float64)
        ownx, owny = x[0] * c.window_width, x[1] * c.window_height
        gx, gy = x[-2] * c.window_width, x[-1] * c.window_height
        dist_goal = math.sqrt((ownx - gx) ** 2 + (owny - gy) ** 2)
        if c.intruder_size != 0:
            for idx in range(c.n):
                ix, iy = x[idx * 4 + 4] * c.window_width, x[idx * 4 + 5] * c.window_height
                di = math.sqrt((ownx - ix) ** 2 + (owny - iy) ** 2)
                if di < self.minimum_separation:
                    return c.NMAC_penalty if di < self.NMAC_dist else c.conflict_penalty
        if dist_goal < self.goal_radius:
            return c.goal_reward
        return c.step_penalty if c.sparse_reward else -dist_goal / 1200


class SingleAircraftDiscrete3HEREnv(SingleAircraftDiscrete9HEREnv):
    """Simulators/SingleAircraftDiscrete3HEREnv.py: three heading actions, goal drawn 100 px inside the map,
    step reward shaped by the nearest intruder; step() returns that distance in place of the info dict (:178)."""
    _variant = "her3"

    def _fetch(self, obs):
        v = self._vec
        if v.cfg.shaped_reward:      # the nearest-intruder distance lands next to no_conflict in host memory
            capi.check(v._lib.uam_nearest(v._h, self._h_side.data_ptr() + 4, v._stream()))
        out = super()._fetch(obs)
        self._nearest = float(self._h_side.numpy()[1:2].view(np.float32)[0])
        return out

    def step(self, action):
        ob, rew, done, _ = self._step_host(action)
        return ob, rew, done, self._nearest


def _her_cfg(config, sparse_reward, knn, num_intruders):
    n = int(config.intruder_size if num_intruders is None else num_intruders)
    cfg = make_config(1, max(n, 1), config=config)
    cfg.num_intruders = n        # 0 skips the intruder test (Config.intruder_size != 0, :256)
    cfg.knn = int(config.n if knn is None else knn)
    if sparse_reward is not None:
        cfg.sparse_reward = int(bool(sparse_reward))
    return cfg


def her_goal_reward(achieved_goal, desired_goal, config=Config, sparse_reward=None):
    """Batched compute_reward (Discrete9HEREnv.py:229-243): float32 device tensors [M, 2] -> reward float32[M]."""
    import torch
    a = achieved_goal.contiguous().to(torch.float32)
    g = desired_goal.contiguous().to(torch.float32)
    if a.shape != g.shape or a.dim() != 2 or a.shape[1] != 2 or not a.is_cuda:
        raise ValueError("achieved_goal / desired_goal must be device tensors of shape [M, 2]")
    out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    cfg = _her_cfg(config, sparse_reward, None, None)
    capi.check(capi.lib().uam_her_goal_reward(C.byref(cfg), a.device.index or 0, a.data_ptr(), g.data_ptr(), a.shape[0],
                                              out.data_ptr(), torch.cuda.current_stream(a.device).cuda_stream))
    return out


def her_input_reward(new_inputs, config=Config, sparse_reward=None, knn=None, num_intruders=None):
    """Batched compute_input_reward (Discrete9HEREnv.py:245-276) for hindsight relabelling
    (Algorithms/pytorch/agent_her.py:105-117): rows observation | goal, float32 device tensor [M, D] ->
    (reward float32[M], done bool[M]) with done = reward is the goal reward or the NMAC penalty."""
    import torch
    x = new_inputs.contiguous().to(torch.float32)
    if x.dim() != 2 or not x.is_cuda:
        raise ValueError("new_inputs must be a device tensor of shape [M, D]")
    rew = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    done = torch.empty((x.shape[0],), dtype=torch.uint8, device=x.device)
    cfg = _her_cfg(config, sparse_reward, knn, num_intruders)
    capi.check(capi.lib().uam_her_input_reward(C.byref(cfg), x.device.index or 0, x.data_ptr(), x.shape[0], x.shape[1],
                                               rew.data_ptr(), done.data_ptr(),
                                               torch.cuda.current_stream(x.device).cuda_stream))
    return rew, done.view(torch.bool)
