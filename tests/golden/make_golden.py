#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE itself.

Runs only in the build container (needs /root/reference).  The reference
simulators are imported unmodified under an in-memory `gym` stub (gym is not
installed; the stub's Box.contains is the gym<=0.12 form the reference was
written against) and driven from np.random.seed(seed) alone.  What
is committed are DATA only: seeds, configs, action lists and the recorded
outputs/state — no reference source.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

Fixture layout (all arrays; T = number of env.step calls):
  meta_*           : scalars (seed, N, variant, config overrides, modes)
  actions          : [T] int32 | [T,2] int32 | [T] float32
  obs0             : [D] observation returned by the first reset()
  obs              : [T, D] float64 observation returned by step() (terminal obs on done)
  reward, done, info, no_conflict : [T]   (info: 0 '' 1 'c' 2 'g' 3 'n'; no_conflict BEFORE any auto-reset)
  reset_steps      : [R] indices t with done -> env.reset() was called right after (DummyVecEnv order)
  reset_obs        : [R, D] the observations those resets returned
  conflict, promoted : [T, N] uint8 per-intruder flags after step t (after the reset when t is in reset_steps)
  st_steps         : [S] steps at which the full state was also snapshotted
  st_<field>       : [S, ...] state after step t (after the auto reset, if any)
  st0_<field>      : state right after the first reset()
  inj_<field>      : (hand-placed scenarios) state injected before the first step
"""
import argparse
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INFO = {"": 0, "c": 1, "g": 2, "n": 3, "w": 4}


# ----------------------------------------------------------------------------- gym stub
def install_gym_stub():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    envs = types.ModuleType("gym.envs")
    cc = types.ModuleType("gym.envs.classic_control")
    rendering = types.ModuleType("gym.envs.classic_control.rendering")
    pyglet = types.ModuleType("pyglet")
    pyglet_gl = types.ModuleType("pyglet.gl")

    class Env:
        pass

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                self.low = np.asarray(low).astype(dtype)
                self.high = np.asarray(high).astype(dtype)
                self.shape = self.low.shape
            else:
                self.low = np.full(shape, low, dtype=dtype)
                self.high = np.full(shape, high, dtype=dtype)
                self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)

        def contains(self, x):  # gym <= 0.12
            return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()

    class Discrete:
        def __init__(self, n):
            self.n = n

        def contains(self, x):
            return 0 <= int(x) < self.n

    class Dict(dict):
        def __init__(self, d):
            super().__init__(d)

    class Geom:
        def __init__(self):
            pass

    def np_random(seed=None):
        return np.random.RandomState(seed), seed

    gym.Env = Env
    gym.GoalEnv = Env
    spaces.Box, spaces.Discrete, spaces.Dict = Box, Discrete, Dict
    seeding.np_random = np_random
    rendering.Geom = Geom
    gym.spaces, gym.utils, gym.envs = spaces, utils, envs
    utils.seeding = seeding
    envs.classic_control = cc
    cc.rendering = rendering
    pyglet.gl = pyglet_gl
    sys.modules.update({
        "gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding,
        "gym.envs": envs, "gym.envs.classic_control": cc, "gym.envs.classic_control.rendering": rendering,
        "pyglet": pyglet, "pyglet.gl": pyglet_gl,
    })
    if not hasattr(np, "float"):
        np.float = float  # SingleAircraft2Env.py:35 uses the removed alias


# ----------------------------------------------------------------------------- helpers
def snapshot(env, raw6=False):
    N = len(env.intruder_list)
    st = {
        "int_x": np.zeros(N), "int_y": np.zeros(N),
        "int_vx": np.zeros(N, np.float32), "int_vy": np.zeros(N, np.float32),
        "int_speed": np.zeros(N), "int_heading": np.zeros(N),
        "int_conflict": np.zeros(N, np.uint8), "int_promoted": np.zeros(N, np.uint8),
    }
    for i, it in enumerate(env.intruder_list):
        st["int_x"][i] = float(it.position[0])
        st["int_y"][i] = float(it.position[1])
        st["int_promoted"][i] = it.position.dtype == np.float64
        assert it.velocity.dtype == np.float32
        st["int_vx"][i] = it.velocity[0]
        st["int_vy"][i] = it.velocity[1]
        st["int_speed"][i] = float(it.speed)
        st["int_heading"][i] = float(it.heading)
        st["int_conflict"][i] = bool(it.conflict)
    d = env.drone
    assert d.position.dtype == np.float32
    st["own_x"] = np.float32(d.position[0])
    st["own_y"] = np.float32(d.position[1])
    st["own_heading"] = np.float64(d.heading)
    st["own_speed"] = np.float64(d.speed)
    st["own_vx"] = np.float64(d.velocity[0])
    st["own_vy"] = np.float64(d.velocity[1])
    st["goal_x"] = np.float64(env.goal.position[0])
    st["goal_y"] = np.float64(env.goal.position[1])
    st["no_conflict"] = np.int32(env.no_conflict)
    return st


def info_code(info):
    if isinstance(info, dict):
        info = info["result"]
    return INFO[info]


def goal_seeking_action(env, d_heading):
    d = env.drone
    bearing = math.atan2(env.goal.position[1] - float(d.position[1]), env.goal.position[0] - float(d.position[0]))
    diff = (bearing - float(d.heading) + math.pi) % (2 * math.pi) - math.pi
    a0 = 2 if diff > d_heading / 2 else (0 if diff < -d_heading / 2 else 1)
    return a0, 2


class Recorder:
    def __init__(self, N, stride):
        self.N, self.stride = N, stride
        self.rows = {k: [] for k in ("obs", "reward", "done", "info", "no_conflict", "conflict", "promoted")}
        self.reset_steps, self.reset_obs = [], []
        self.st_steps, self.st = [], {}

    def add_state(self, t, st):
        self.st_steps.append(t)
        for k, v in st.items():
            self.st.setdefault(k, []).append(v)

    def finish(self):
        out = {k: np.asarray(v) for k, v in self.rows.items()}
        out["reward"] = out["reward"].astype(np.float64)
        out["done"] = out["done"].astype(np.uint8)
        out["info"] = out["info"].astype(np.uint8)
        out["no_conflict"] = out["no_conflict"].astype(np.int32)
        out["reset_steps"] = np.asarray(self.reset_steps, dtype=np.int64)
        D = out["obs"].shape[1]
        out["reset_obs"] = np.asarray(self.reset_obs, dtype=np.float64).reshape(len(self.reset_steps), D)
        out["st_steps"] = np.asarray(self.st_steps, dtype=np.int64)
        for k, v in self.st.items():
            out["st_" + k] = np.asarray(v)
        return out


def run_episode_stream(env, actions_fn, T, stride, step_fn=None, post_step=None):
    """DummyVecEnv-order driver: step, and reset right after a done."""
    N = len(env.intruder_list)
    rec = Recorder(N, stride)
    acts = []
    for t in range(T):
        a = actions_fn(t, env)
        acts.append(a)
        obs, r, d, info = (step_fn or (lambda e, x: e.step(x)))(env, a)
        rec.rows["obs"].append(np.asarray(obs, dtype=np.float64))
        rec.rows["reward"].append(float(r))
        rec.rows["done"].append(bool(d))
        rec.rows["info"].append(info_code(info))
        rec.rows["no_conflict"].append(int(env.no_conflict))
        if d:
            ro = env.reset()
            rec.reset_steps.append(t)
            rec.reset_obs.append(np.asarray(ro, dtype=np.float64))
        st = snapshot(env)
        rec.rows["conflict"].append(st["int_conflict"])
        rec.rows["promoted"].append(st["int_promoted"])
        if t % stride == 0 or d or t == T - 1 or t < 8:
            rec.add_state(t, st)
    out = rec.finish()
    out["actions"] = np.asarray(acts)
    return out


def save(name, meta, data):
    path = os.path.join(HERE, name + ".npz")
    payload = {("meta_" + k): np.asarray(v) for k, v in meta.items()}
    payload.update(data)
    np.savez_compressed(path, **payload)
    codes = data.get("info")
    summ = "" if codes is None else " codes=" + str(np.bincount(np.asarray(codes).ravel(), minlength=5).tolist())
    prom = data.get("promoted")
    if prom is not None:
        summ += " promoted_frac=%.4f" % float(np.mean(prom))
    print("wrote %-28s %7.1f KB%s" % (name + ".npz", os.path.getsize(path) / 1024, summ))


def add_prefixed(data, prefix, st):
    for k, v in st.items():
        data[prefix + k] = np.asarray(v)


# ----------------------------------------------------------------------------- scenarios
def load_reference(ref):
    install_gym_stub()
    sys.path.insert(0, os.path.join(ref, "Simulators"))
    import config  # noqa: F401  (reference's Simulators/config.py)
    import SingleAircraftEnv as m_env
    import SingleAircraft2Env as m_env2
    import SingleAircraftMCTSEnv as m_mcts
    import SingleAircraftMCTSRandIntruderEnv as m_ri
    return config.Config, m_env, m_env2, m_mcts, m_ri


def g1_cyclic(Config, m_env):
    """G1: discrete-9, N=4, seed 0, 512 cyclic actions (BASELINE config 1)."""
    Config.intruder_size = 4
    np.random.seed(0)
    env = m_env.SingleAircraftEnv()
    obs0 = env.reset()
    st0 = snapshot(env)
    data = run_episode_stream(env, lambda t, e: t % 9, 512, 16)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = data["actions"].astype(np.int32)
    save("g1_discrete9_n4_seed0", dict(seed=0, N=4, variant="SingleAircraftEnv", obs_mode=0, action_mode=0,
                                       dynamics=0), data)


def g2_goal_seeking(Config, m_env, N, seed, T):
    """G2: goal-seeking policy so that 'g', 'n' and 'c' all occur."""
    Config.intruder_size = N
    np.random.seed(seed)
    env = m_env.SingleAircraftEnv()
    obs0 = env.reset()
    st0 = snapshot(env)

    def act(t, e):
        a0, a1 = goal_seeking_action(e, Config.d_heading)
        return a0 * 3 + a1

    data = run_episode_stream(env, act, T, 25)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = data["actions"].astype(np.int32)
    save("g2_goalseek_n%d_seed%d" % (N, seed), dict(seed=seed, N=N, variant="SingleAircraftEnv", obs_mode=0,
                                                    action_mode=0, dynamics=0), data)


def hand_placed(Config, m_env, name, N, seed, place_fn, T, doc):
    """G3-G5: sigmas zeroed, state edited after reset(), then T no-op steps."""
    keep = (Config.heading_sigma, Config.speed_sigma)
    Config.heading_sigma, Config.speed_sigma = 0.0, 0.0
    try:
        Config.intruder_size = N
        np.random.seed(seed)
        env = m_env.SingleAircraftEnv()
        obs0 = env.reset()
        place_fn(env)
        inj = snapshot(env)
        data = run_episode_stream(env, lambda t, e: 4, T, 1)
        data["obs0"] = np.asarray(obs0)
        add_prefixed(data, "inj_", inj)
        data["actions"] = data["actions"].astype(np.int32)
        save(name, dict(seed=seed, N=N, variant="SingleAircraftEnv", obs_mode=0, action_mode=0, dynamics=0,
                        heading_sigma=0.0, speed_sigma=0.0, doc=doc), data)
    finally:
        Config.heading_sigma, Config.speed_sigma = keep


def own_next(env):
    d = env.drone
    vx = d.speed * math.cos(d.heading)
    vy = d.speed * math.sin(d.heading)
    p = d.position.copy()
    p += np.array([vx, vy])
    return p  # float32


def place_g3(env):
    """NMAC on intruder k=3; intruder 1 in conflict range; intruders 5 and 6 (> k) also inside NMAC /
    about to leave the map: they must stay untouched by the early exit."""
    nxt = own_next(env)
    il = env.intruder_list
    k = 3
    il[k].position = (nxt - il[k].velocity + np.array([1.0, 0.5], np.float32)).astype(np.float32)
    il[1].position = (nxt - il[1].velocity + np.array([-9.0, 4.0], np.float32)).astype(np.float32)
    il[5].position = (nxt - il[5].velocity + np.array([0.5, -1.0], np.float32)).astype(np.float32)
    il[6].position = np.array([799.9, 799.9], np.float32)
    il[6].velocity = np.array([1.5, 1.5], np.float32)


def place_g4(env):
    """Out-of-map respawn coinciding with a conflict: old distance is used, no_conflict still
    increments, the replacement's flag is clear."""
    d = env.drone
    d.position = np.array([790.0, 400.0], np.float32)
    d.heading = 0.0
    il = env.intruder_list
    il[2].position = np.array([799.5, 400.0], np.float32)
    il[2].velocity = np.array([2.0, 0.0], np.float32)
    # and one flagged intruder staying in conflict: must not increment again
    il[0].position = np.array([780.0, 410.0], np.float32)
    il[0].velocity = np.array([1.6, 0.0], np.float32)
    il[0].conflict = True
    env.goal.position = np.array([100.0, 100.0])


def place_g5(env):
    """Inside the goal radius while in conflict => 'c', not done; then the intruder leaves => 'g'."""
    nxt = own_next(env)
    env.goal.position = np.array([float(nxt[0]) + 5.0, float(nxt[1]) + 2.0])
    il = env.intruder_list
    il[0].position = (nxt + np.array([12.0, 0.0], np.float32)).astype(np.float32)
    il[0].velocity = np.array([2.6, 0.0], np.float32)
    d = env.drone
    d.speed = 0.0  # stays put (clamped back up to min_speed in step(); still inside the radius for a few steps)


def g7_rng_kat():
    data = {}
    for s in (0, 2, 12345):
        np.random.seed(s)
        data["random_sample_%d" % s] = np.random.random_sample(64)
        np.random.seed(s)
        data["normal_%d" % s] = np.array([np.random.normal(0, 0.5) for _ in range(64)])
        np.random.seed(s)
        np.random.random_sample(700)  # crosses one twist
        st = np.random.get_state()
        data["state_key_%d" % s] = st[1].astype(np.uint32)
        data["state_pos_%d" % s] = np.int32(st[2])
        np.random.seed(s)
        mix = []
        for i in range(300):  # mixed draws: uniform pairs, normals (cache!), scalars
            mix.extend(np.random.uniform(low=np.array([0, 0]), high=np.array([800, 800])).tolist())
            mix.append(np.random.normal(0, 0.07))
            mix.append(np.random.uniform(low=5 / 3, high=8 / 3))
        data["mixed_%d" % s] = np.asarray(mix)
    save("g7_rng_kat", dict(seeds=[0, 2, 12345]), data)


def g8_continuous(Config, m_env2):
    """G8: SingleAircraft2Env dynamics (Simulators semantics: heading += d_heading*(a-1), no speed
    command).  Driven through drone.step/_terminal_reward/_get_ob because step()'s Box(1,) assert
    forces an array action that breaks _get_ob on modern NumPy."""
    N = 8
    Config.intruder_size = N
    np.random.seed(7)
    env = m_env2.SingleAircraft2Env()
    obs0 = env.reset()
    st0 = snapshot(env)
    arng = np.random.RandomState(99)  # private stream: does not touch the global one
    acts = arng.uniform(-1, 1, size=600).astype(np.float32)

    def step_fn(e, a):
        e.drone.step(float(a))
        reward, terminal, info = e._terminal_reward()
        return e._get_ob(), reward, terminal, {"result": info}

    data = run_episode_stream(env, lambda t, e: acts[t], 600, 20, step_fn=step_fn)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = acts
    save("g8_continuous_n8_seed7", dict(seed=7, N=N, variant="SingleAircraft2Env", obs_mode=0, action_mode=2,
                                        dynamics=0), data)


def g8b_packaged_continuous2(ref, N=8, seed=17, name="g8b_packaged_continuous2_n8_seed17", T=900):
    """G8 'second mode': the PACKAGED copy gym_guidance_collision_avoidance_single/envs/SingleAircraft2Env.py — action
    (a[0], a[1]) un-offset with a live speed command (:281-285), its own reward literals (NMAC -5, conflict -1,
    goal +1) and the out-of-map termination 'w' (-100, :169-170), heading_sigma 2 degrees (envs/config.py).  The module
    uses a relative import (`from .config import Config`): it is loaded under a throw-away package name."""
    import importlib
    import types
    pkg_dir = os.path.join(ref, "gym_guidance_collision_avoidance_single", "envs")
    pkg = types.ModuleType("_gcas_packaged_envs")
    pkg.__path__ = [pkg_dir]
    sys.modules["_gcas_packaged_envs"] = pkg
    m = importlib.import_module("_gcas_packaged_envs.SingleAircraft2Env")
    PConfig = importlib.import_module("_gcas_packaged_envs.config").Config
    PConfig.intruder_size = N      # (N = 0 is the packaged config's own default: a map with the ownship and a goal only)
    np.random.seed(seed)
    env = m.SingleAircraft2Env()
    obs0 = env.reset()
    st0 = snapshot(env)
    arng = np.random.RandomState(98)   # private stream for the action noise
    episodes = [0]
    real_reset = env.reset

    def counting_reset():
        episodes[0] += 1
        return real_reset()
    env.reset = counting_reset

    def act(t, e):
        d = e.drone
        x, y = float(d.position[0]), float(d.position[1])
        if episodes[0] % 2 == 1:                      # every other episode heads for the nearest map edge: the 'w' rule
            tx, ty = (x, -50.0) if y < x else (-50.0, y)
        else:
            tx, ty = e.goal.position[0], e.goal.position[1]
        bearing = math.atan2(ty - y, tx - x)
        diff = (bearing - float(d.heading) + math.pi) % (2 * math.pi) - math.pi
        a0 = np.clip(diff / PConfig.d_heading + arng.normal(0, 0.3), -1, 1)
        a1 = np.clip(arng.uniform(-1, 1), -1, 1)
        return np.array([np.float32(a0), np.float32(a1)], dtype=np.float64)   # float32-valued, like the C-ABI carries

    data = run_episode_stream(env, act, T, 20)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = np.asarray(data["actions"], dtype=np.float32)
    save(name,
         dict(seed=seed, N=N, variant="packaged SingleAircraft2Env", obs_mode=0, action_mode=4, dynamics=0,
              heading_sigma=PConfig.heading_sigma, speed_sigma=PConfig.speed_sigma, nmac_penalty=-5.0,
              conflict_penalty=-1.0, wall_penalty=-100.0, goal_reward=1.0, wall_terminates=1), data)


def g9_randintruder(Config, m_ri, N, seed, T):
    """G9: RandIntruder env: constant drift, Bernoulli heading changes (also on terminal steps), raw
    6-wide obs, pair actions."""
    Config.intruder_size = N
    np.random.seed(seed)
    env = m_ri.SingleAircraftEnv()
    obs0 = env.reset()
    st0 = snapshot(env)

    def act(t, e):
        return goal_seeking_action(e, Config.d_heading)

    data = run_episode_stream(env, act, T, 20)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = data["actions"].astype(np.int32).reshape(T, 2)
    save("g9_randintruder_n%d_seed%d" % (N, seed), dict(seed=seed, N=N, variant="SingleAircraftMCTSRandIntruderEnv",
                                                        obs_mode=2, action_mode=1, dynamics=1), data)


def g9b_mctsenv(Config, m_mcts):
    """SingleAircraftMCTSEnv: raw 4-wide obs, pair action."""
    N = 8
    Config.intruder_size = N
    np.random.seed(11)
    env = m_mcts.SingleAircraftEnv()
    obs0 = env.reset()
    st0 = snapshot(env)
    data = run_episode_stream(env, lambda t, e: goal_seeking_action(e, Config.d_heading), 800, 20)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = data["actions"].astype(np.int32).reshape(800, 2)
    save("g9b_mctsenv_n8_seed11", dict(seed=11, N=N, variant="SingleAircraftMCTSEnv", obs_mode=1, action_mode=1,
                                       dynamics=0), data)


def g10_shared_stream(Config, m_env):
    """G10: 4 env objects in one process sharing the global np.random, stepped in DummyVecEnv order."""
    N, B, T = 8, 4, 400
    Config.intruder_size = N
    np.random.seed(5)
    envs = [m_env.SingleAircraftEnv() for _ in range(B)]
    obs0 = np.stack([e.reset() for e in envs])
    D = obs0.shape[1]
    obs = np.zeros((T, B, D))
    rew = np.zeros((T, B))
    done = np.zeros((T, B), np.uint8)
    info = np.zeros((T, B), np.uint8)
    nc = np.zeros((T, B), np.int32)
    acts = np.zeros((T, B), np.int32)
    for t in range(T):
        for b, e in enumerate(envs):
            a0, a1 = goal_seeking_action(e, Config.d_heading)
            a = a0 * 3 + a1
            acts[t, b] = a
            o, r, d, inf = e.step(a)
            rew[t, b], done[t, b], info[t, b], nc[t, b] = r, d, info_code(inf), e.no_conflict
            if d:
                o = e.reset()  # VecEnv semantics: obs of a done env is its reset obs
            obs[t, b] = o
    save("g10_shared_stream_b4_n8_seed5", dict(seed=5, N=N, B=B, variant="SingleAircraftEnv", obs_mode=0,
                                               action_mode=0, dynamics=0),
         dict(obs0=obs0, obs_after=obs, reward=rew, done=done, info=info, no_conflict=nc, actions=acts))


def g6_randomenv(Config, N, seed, T):
    """G6b: SingleAircraftRandomEnv (random ownship start), goal-seeking policy."""
    import SingleAircraftRandomEnv as m_rand
    Config.intruder_size = N
    np.random.seed(seed)
    env = m_rand.SingleAircraftRandomEnv()
    obs0 = env.reset()
    st0 = snapshot(env)

    def act(t, e):
        a0, a1 = goal_seeking_action(e, Config.d_heading)
        return a0 * 3 + a1

    data = run_episode_stream(env, act, T, 25)
    data["obs0"] = np.asarray(obs0)
    add_prefixed(data, "st0_", st0)
    data["actions"] = data["actions"].astype(np.int32)
    save("g6b_randomenv_n%d_seed%d" % (N, seed), dict(seed=seed, N=N, variant="SingleAircraftRandomEnv", obs_mode=0,
                                                     action_mode=0, dynamics=0, random_own_start=1), data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    Config, m_env, m_env2, m_mcts, m_ri = load_reference(args.ref)
    g7_rng_kat()
    g1_cyclic(Config, m_env)
    g2_goal_seeking(Config, m_env, 8, 3, 3000)
    g2_goal_seeking(Config, m_env, 32, 1, 2500)
    g2_goal_seeking(Config, m_env, 80, 4, 600)   # shipped Config.intruder_size: two 64-bit mask words
    hand_placed(Config, m_env, "g3_nmac_early_exit", 8, 21, place_g3, 3,
                "NMAC at idx 3: intruders 4.. are not integrated / respawned / flagged")
    hand_placed(Config, m_env, "g4_respawn_in_conflict", 8, 22, place_g4, 4,
                "respawn coinciding with conflict; flagged intruder does not re-count")
    hand_placed(Config, m_env, "g5_conflict_masks_goal", 8, 23, place_g5, 12,
                "conflict masks goal until the intruder leaves")
    g8_continuous(Config, m_env2)
    g8b_packaged_continuous2(args.ref)
    g8b_packaged_continuous2(args.ref, N=0, seed=23, name="g8c_packaged_default_n0_seed23", T=1500)
    g9_randintruder(Config, m_ri, 8, 13, 1500)
    g9_randintruder(Config, m_ri, 64, 14, 500)
    g9b_mctsenv(Config, m_mcts)
    g10_shared_stream(Config, m_env)
    g6_randomenv(Config, 8, 31, 2000)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g8c":       # only the packaged default (no intruders) fixture
        sys.dont_write_bytecode = True
        install_gym_stub()
        g8b_packaged_continuous2("/root/reference", N=0, seed=23, name="g8c_packaged_default_n0_seed23", T=1500)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g8b":       # only the packaged-copy fixture
        sys.dont_write_bytecode = True
        install_gym_stub()
        g8b_packaged_continuous2("/root/reference")
    else:
        main()
