"""Shared harness for the GPU parity tests: drive the HIP env (through the C-ABI host layer) and the
CPU oracle with identical configs, seeds and actions, and compare them step by step."""
import numpy as np

STATE_EXACT = ["int_x", "int_y", "int_vx", "int_vy", "int_conflict", "int_promoted", "own_x", "own_y",
               "own_heading", "own_speed", "own_vx", "own_vy", "goal_x", "goal_y", "no_conflict", "reset_count", "step_index"]


def oracle_cfg_from(oracle, cfg):
    """Copy a product `uam_config` into the oracle's mirror struct field by field."""
    ocfg = oracle.UamConfig()
    for name, _ in oracle.UamConfig._fields_:
        setattr(ocfg, name, getattr(cfg, name))
    return ocfg


def goal_seek_actions(st, d_heading, action_mode):
    """The heuristic used for the golden fixtures: turn toward the goal, full throttle."""
    bearing = np.arctan2(st["goal_y"] - st["own_y"].astype(np.float64), st["goal_x"] - st["own_x"].astype(np.float64))
    diff = (bearing - st["own_heading"] + np.pi) % (2 * np.pi) - np.pi
    a0 = np.where(diff > d_heading / 2, 2, np.where(diff < -d_heading / 2, 0, 1)).astype(np.int32)
    a1 = np.full_like(a0, 2)
    if action_mode == 0:
        return a0 * 3 + a1
    if action_mode == 1:
        return np.stack([a0, a1], axis=1).astype(np.int32)
    if action_mode == 3:
        return a0
    if action_mode == 4:      # packaged 2Env: un-offset heading command, full throttle
        return np.stack([np.clip(diff / d_heading, -1, 1), np.ones_like(diff)], axis=1).astype(np.float32)
    return np.clip(1.0 + diff / d_heading, -1, 1).astype(np.float32)


def random_actions(rs, B, action_mode):
    if action_mode == 0:
        return rs.randint(0, 9, size=B).astype(np.int32)
    if action_mode == 1:
        return rs.randint(0, 3, size=(B, 2)).astype(np.int32)
    if action_mode == 3:
        return rs.randint(0, 3, size=B).astype(np.int32)
    if action_mode == 4:
        return rs.uniform(-1, 1, size=(B, 2)).astype(np.float32)
    return rs.uniform(-1, 1, size=B).astype(np.float32)


def compare_states(gs, os_, where, fields=None, hold_sh=False):
    fields = list(fields or STATE_EXACT)
    if hold_sh:
        fields += ["int_speed", "int_heading"]
    for k in fields:
        if k not in gs or k not in os_:
            continue
        a, b = np.atleast_1d(gs[k]), np.atleast_1d(os_[k]).astype(np.asarray(gs[k]).dtype)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            i = tuple(bad[0])
            raise AssertionError("%s: state field %s differs in %d places, first at %s: gpu=%r oracle=%r" %
                                 (where, k, len(bad), i, a[i], b[i]))


def compare_outputs(g, o, where, obs_tol, reward_tol=0.0):
    gobs, grew, gdone, ginfo = g
    oobs, orew, odone, oinfo = o
    assert np.array_equal(gdone.astype(np.uint8), odone), where + ": done flags differ at %s" % np.argwhere(gdone != odone)[:4].tolist()
    assert np.array_equal(ginfo, oinfo), where + ": info codes differ at %s" % np.argwhere(ginfo != oinfo)[:4].tolist()
    if reward_tol:
        assert (np.abs(grew - orew) <= reward_tol).all(), where + ": rewards differ (max |d| = %g)" % np.abs(grew - orew).max()
    else:
        assert np.array_equal(grew, orew.astype(np.float32)), where + ": rewards differ (max |d| = %g)" % np.abs(grew - orew).max()
    if gobs is not None:
        d = np.abs(gobs.astype(np.float64) - oobs)
        lim = obs_tol * np.maximum(1.0, np.abs(oobs))
        assert (d <= lim).all(), where + ": obs differs, max |d| = %g at %s" % (d.max(), np.unravel_index(d.argmax(), d.shape))


def run_lockstep(uamenv, oracle, B, N, T, policy="random", seed=123, check_state_every=1, obs_tol=1e-6,
                 act_seed=7, reward_tol=0.0, **cfg_kw):
    """Step GPU env and oracle in lockstep for T steps; every comparison is against the oracle on the SAME
    inputs.  Returns summary counters (dones per code, spawns) so tests can assert coverage."""
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=seed, **cfg_kw)
    ocfg = oracle_cfg_from(oracle, env.cfg)
    orc = oracle.OracleBatch(ocfg, oracle.MATH_PORTABLE, threads=8)
    hold_sh = env.cfg.dynamics == 1 or env.cfg.obs_mode == 2
    am = env.cfg.action_mode
    gobs = env.reset().cpu().numpy()
    oobs = orc.reset()
    d = np.abs(gobs.astype(np.float64) - oobs)
    assert (d <= obs_tol * np.maximum(1.0, np.abs(oobs))).all(), "reset obs differs: %g" % d.max()
    compare_states(env.get_state(), orc.get_state(), "after reset", hold_sh=hold_sh)
    rs = np.random.RandomState(act_seed)
    counts = np.zeros(5, dtype=np.int64)
    ost = orc.get_state()
    for t in range(T):
        if policy == "random":
            acts = random_actions(rs, B, am)
        elif policy == "goal":
            acts = goal_seek_actions(ost, env.cfg.d_heading, am)
        else:  # mixed: goal seeking with 20 % random actions
            acts = goal_seek_actions(ost, env.cfg.d_heading, am)
            r = random_actions(rs, B, am)
            m = rs.uniform(size=B) < 0.2
            acts[m] = r[m]
        o, r_, dn, inf = env.step(acts)
        g = (o.cpu().numpy(), r_.cpu().numpy(), dn.cpu().numpy(), inf.codes.cpu().numpy())
        oo = orc.step(acts)
        where = "step %d" % t
        compare_outputs(g, oo, where, obs_tol, reward_tol)
        if env.cfg.shaped_reward:     # Discrete3HEREnv.step's 4th return value
            gn, (on, _) = env.nearest().cpu().numpy(), orc.nearest()
            assert (np.abs(gn - on) <= 1e-6 * np.maximum(1.0, on)).all(), where + ": nearest-intruder distance differs"
        counts += np.bincount(oo[3], minlength=5)
        ost = orc.get_state()
        if check_state_every and (t % check_state_every == 0 or t == T - 1):
            compare_states(env.get_state(), ost, where, hold_sh=hold_sh)
    out = dict(counts=counts, spawns=orc.spawn_count, redraws=orc.redraw_count)
    env.close()
    orc.close()
    return out
