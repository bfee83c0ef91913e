"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs, against the committed reference-generated golden fixtures, and — at BASELINE.json's full
sizes — through size-independent properties.

Bars (north_star): discrete collision/done/info flags bit-exact; float32 state bit-exact against the
oracle built with the same portable math; rewards bit-exact as float32; observations within 1e-6
(the float32 pack rounds the reference's float64 tail entries).  Against the reference's own
trajectories (golden fixtures, libm trig): flags exact, floats within 1e-5 abs (north_star's bound).
"""
import numpy as np
import pytest

import golden_util as gu
import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uamenv():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import uamenv as m
    m._capi.lib()   # raises if the HIP extension is missing: no fallback
    return m


# ---------------------------------------------------------------- Philox (throughput mode) vs oracle
@pytest.mark.parametrize("B,N,T,policy", [
    (256, 4, 300, "goal"),      # L=1: one lane per env
    (512, 8, 400, "mixed"),     # BASELINE config 2 shape (smaller batch)
    (384, 32, 400, "goal"),     # BASELINE config 3 shape: 8 lanes per env
    (96, 80, 250, "goal"),      # shipped Config.intruder_size: 20 of 32 lanes active
    (200, 5, 200, "goal"),      # N not a multiple of 4: padded quad, scalar obs stores
    (64, 20, 200, "mixed"),     # MCTS config N
    (33, 256, 60, "goal"),      # one full wave per env
    (4096, 8, 64, "random"),    # BASELINE config 2 at full batch
])
def test_philox_f32_matches_oracle(uamenv, oracle, B, N, T, policy):
    out = pu.run_lockstep(uamenv, oracle, B, N, T, policy=policy, rng="philox", precision="f32",
                          check_state_every=10)
    if policy != "random" and B * T >= 10000:
        assert out["counts"][1] > 0, "no conflict step exercised: %s" % out["counts"]
        assert out["spawns"] > B * N, "no respawn exercised"


@pytest.mark.parametrize("B,N,T", [(256, 8, 300), (128, 32, 300), (64, 80, 150)])
def test_philox_exact_promotion_matches_oracle(uamenv, oracle, B, N, T):
    """float64 'promoted' positions after a rejection redraw (SingleAircraftEnv.py:85-86,236-237)."""
    out = pu.run_lockstep(uamenv, oracle, B, N, T, policy="goal", rng="philox", precision="exact",
                          check_state_every=10)
    assert out["redraws"] > 0, "no promotion exercised"


@pytest.mark.parametrize("variant,N", [("continuous", 8), ("continuous2", 32), ("mcts", 8), ("randintruder", 8), ("randintruder", 64),
                                       ("random_start", 8)])
@pytest.mark.parametrize("precision", ["f32", "exact"])
def test_variants_match_oracle(uamenv, oracle, variant, N, precision):
    pu.run_lockstep(uamenv, oracle, 192, N, 200, policy="goal", variant=variant, rng="philox", precision=precision,
                    check_state_every=10)


def test_sparse_reward_and_no_auto_reset(uamenv, oracle):
    class Cfg(uamenv.Config):
        sparse_reward = True
    pu.run_lockstep(uamenv, oracle, 128, 8, 120, policy="goal", rng="philox", precision="f32", config=Cfg,
                    auto_reset=False, check_state_every=10)


# ---------------------------------------------------------------- MT19937 (NumPy stream) vs oracle
@pytest.mark.parametrize("variant,N,T", [("discrete9", 8, 400), ("discrete9", 32, 300), ("randintruder", 8, 200),
                                         ("continuous", 8, 150)])
def test_mt19937_matches_oracle(uamenv, oracle, variant, N, T):
    """One NumPy-legacy stream per env, consumed in the reference's draw order; oracle built with the same
    portable math, so even the float state must match to the bit."""
    pu.run_lockstep(uamenv, oracle, 96, N, T, policy="goal", variant=variant, rng="mt19937", precision="exact",
                    check_state_every=10)


def test_mt19937_full_wave_envs_at_full_block_size(uamenv, oracle):
    """N = 256 (one wave per env, four intruders per reset worker) with a batch large enough that the block keeps its
    256 threads: the per-wave MT19937 reset areas then put the block's dynamic LDS above 64 KB (76.7 KB), the largest
    any configuration asks for; uam_create checks it against the device's limit.  Random actions fly the ownship out of
    the goal corridor slowly, so resets come from NMACs in the dense field."""
    out = pu.run_lockstep(uamenv, oracle, 4096, 256, 12, policy="random", rng="mt19937", precision="exact",
                          check_state_every=4)
    assert out["spawns"] > 4096 * 256, "no respawn exercised"
    assert out["counts"][2:].sum() > 0, "no episode end exercised: %s" % out["counts"]


# ---------------------------------------------------------------- against the reference's own trajectories
VARIANT_OF = {(0, 0, 0): "discrete9", (0, 2, 0): "continuous", (1, 1, 0): "mcts", (2, 1, 1): "randintruder",
              (0, 4, 0): "continuous2"}


@pytest.mark.parametrize("name", gu.TRAJECTORY_FIXTURES)
def test_gpu_replays_reference_fixture_from_seed(uamenv, name):
    """The HIP env, seeded like np.random.seed(s), replays what the REFERENCE produced: flags/counters
    bit-exact, floats within north_star's 1e-5 (device trig/log differ from glibc by <= 1 ulp)."""
    fx = gu.load(name)
    N = int(fx["meta_N"])
    variant = VARIANT_OF[(int(fx["meta_obs_mode"]), int(fx["meta_action_mode"]), int(fx["meta_dynamics"]))]
    if int(fx.get("meta_random_own_start", 0)):
        variant = "random_start"          # Simulators/SingleAircraftRandomEnv.py
    over = {k: float(fx["meta_" + k]) for k in ("heading_sigma", "speed_sigma", "nmac_penalty", "conflict_penalty",
                                                "wall_penalty", "goal_reward") if "meta_" + k in fx}
    if "meta_wall_terminates" in fx:
        over["wall_terminates"] = int(fx["meta_wall_terminates"])
    env = uamenv.BatchedSingleAircraftEnv(1, N, variant=variant, rng="mt19937", precision="exact",
                                          seed=int(fx["meta_seed"]), auto_reset=False, **over)
    raw = int(fx["meta_obs_mode"]) != 0
    tol = 1e-5

    def close(a, b, what):
        """north_star's 1e-5 abs on normalised observations; RAW observations carry positions up to 800 whose float32
        ulp is 6.1e-5, so there the bound is ONE float32 ulp of the value (device trig / log vs glibc)."""
        d = np.abs(a.astype(np.float64) - b)
        lim = np.maximum(np.spacing(np.abs(b).astype(np.float32)).astype(np.float64), 1e-7) if raw else tol
        assert (d <= lim).all(), "%s: max |d| = %g (limit %g)" % (what, d.max(), np.max(lim))

    st_index = {int(t): i for i, t in enumerate(fx["st_steps"])} if "st_steps" in fx else {}
    hold_sh = variant == "randintruder"

    def check_snapshot(t, where):
        """the fixture's SoA snapshot after step t against uam_get_state: flags and counters exact, float32 fields
        within one float32 ulp, float64 fields to 1e-9 relative (device trig vs glibc, never accumulated bits)"""
        if t not in st_index:
            return
        want = gu.prefixed_state(fx, "st_", st_index[t])
        got = env.get_state([k for k in want if k in ("int_x", "int_y", "int_vx", "int_vy", "int_conflict", "int_promoted",
                                                      "own_x", "own_y", "own_heading", "own_speed", "own_vx", "own_vy",
                                                      "goal_x", "goal_y", "no_conflict") or (hold_sh and k in ("int_speed", "int_heading"))])
        for k, g in got.items():
            if k == "step_index":
                continue
            w = want[k]
            if g.dtype.kind in "ui":
                assert np.array_equal(g, w.astype(g.dtype)), "%s: state %s" % (where, k)
            elif k in ("int_x", "int_y", "int_vx", "int_vy", "own_x", "own_y"):
                prom = want["int_promoted"].astype(bool) if k in ("int_x", "int_y") else np.zeros(w.shape, bool)
                ulp = np.where(prom, 1e-9 * np.maximum(1.0, np.abs(w)), np.spacing(np.abs(w).astype(np.float32)).astype(np.float64))
                assert (np.abs(g.astype(np.float64) - w) <= np.maximum(ulp, 1e-9)).all(), "%s: state %s, max |d| = %g" % (
                    where, k, np.abs(g.astype(np.float64) - w).max())
            else:
                assert (np.abs(g - w) <= 1e-9 * np.maximum(1.0, np.abs(w))).all(), "%s: state %s" % (where, k)

    close(env.reset().cpu().numpy()[0], fx["obs0"], name + " reset obs")
    inj = gu.prefixed_state(fx, "inj_")
    if inj:
        hold = {"int_speed", "int_heading"}
        env.set_state({k: v for k, v in inj.items() if k not in hold})
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    actions = fx["actions"]
    for t in range(len(actions)):
        obs, rew, done, infos = env.step(actions[t:t + 1])
        where = "%s step %d" % (name, t)
        assert bool(done[0].item()) == bool(fx["done"][t]), where + ": done"
        assert int(infos.codes[0].item()) == int(fx["info"][t]), where + ": info"
        assert abs(float(rew[0].item()) - fx["reward"][t]) <= tol, where + ": reward"
        close(obs.cpu().numpy()[0], fx["obs"][t], where + " obs")
        assert int(env.no_conflict[0].item()) == int(fx["no_conflict"][t]), where + ": no_conflict"
        if fx["done"][t]:
            close(env.reset().cpu().numpy()[0], fx["reset_obs"][reset_steps[t]], where + " reset obs")
        check_snapshot(t, where)           # the fixtures snapshot the state AFTER the reset that follows a done step
        if t % 16 == 0 or fx["done"][t]:
            st = env.get_state(["int_conflict", "int_promoted"])
            assert np.array_equal(st["int_conflict"][0], fx["conflict"][t]), where + ": conflict flags"
            assert np.array_equal(st["int_promoted"][0], fx["promoted"][t]), where + ": promoted flags"
    env.close()


@pytest.mark.parametrize("name", ["g3_nmac_early_exit", "g4_respawn_in_conflict", "g5_conflict_masks_goal"])
def test_benchmarked_instance_replays_deterministic_reference_fixtures(uamenv, name):
    """The instance bench.py times (Philox, float32 positions, lean spawn) against the REFERENCE directly, no oracle in
    between: G3 / G4 / G5 are deterministic (sigmas zero, state injected after reset(), no promoted intruder), so up to
    the first episode end nothing depends on the random stream except the slot a respawn refills.  done / info /
    no_conflict / conflict flags bit-exact, float32 intruder state bit-exact, observation and reward <= 1e-5."""
    fx = gu.load(name)
    N = int(fx["meta_N"])
    assert not fx["inj_int_promoted"].any()
    env = uamenv.BatchedSingleAircraftEnv(1, N, variant="discrete9", rng="philox", precision="f32", seed=5,
                                          auto_reset=False, heading_sigma=float(fx["meta_heading_sigma"]),
                                          speed_sigma=float(fx["meta_speed_sigma"]))
    env.reset()
    inj = gu.prefixed_state(fx, "inj_")
    env.set_state({k: v for k, v in inj.items() if k not in ("int_speed", "int_heading")})
    prev_vx = inj["int_vx"][0]
    respawned = np.zeros(N, bool)          # slots the reference refilled from ITS stream: not comparable afterwards
    steps = 0
    for t in range(len(fx["actions"])):
        obs, rew, done, infos = env.step(fx["actions"][t:t + 1])
        where = "%s step %d" % (name, t)
        if not fx["done"][t]:
            st_vx = fx["st_int_vx"][t]
            respawned |= st_vx != prev_vx
            prev_vx = st_vx
        assert bool(done[0].item()) == bool(fx["done"][t]), where + ": done"
        assert int(infos.codes[0].item()) == int(fx["info"][t]), where + ": info"
        assert int(env.no_conflict[0].item()) == int(fx["no_conflict"][t]), where + ": no_conflict"
        assert abs(float(rew[0].item()) - fx["reward"][t]) <= 1e-5, where + ": reward"
        st = env.get_state(["int_x", "int_y", "int_vx", "int_vy", "int_conflict", "int_promoted", "own_x", "own_y"])
        if not fx["done"][t]:
            assert np.array_equal(st["int_conflict"][0], fx["conflict"][t]), where + ": conflict flags"
        assert not st["int_promoted"].any()
        keep = ~respawned
        if not fx["done"][t]:              # (a done step's snapshot is the reference's state after ITS reset)
            for k in ("int_x", "int_y", "int_vx", "int_vy"):
                assert np.array_equal(st[k][0][keep].astype(np.float32), fx["st_" + k][t][keep].astype(np.float32)), where + ": " + k
            for k in ("own_x", "own_y"):  # float32(float64 position + fp64 velocity): device sincos vs glibc, <= 1 ulp
                assert abs(float(st[k][0]) - float(fx["st_" + k][t])) <= float(np.spacing(np.float32(abs(fx["st_" + k][t])))), where + ": " + k
        o = obs.cpu().numpy()[0].astype(np.float64)
        mask = np.ones(4 * N + 8, bool)
        for i in np.nonzero(respawned)[0]:
            mask[4 * i:4 * i + 4] = False
        assert np.abs(o - fx["obs"][t])[mask].max() <= 1e-5, where + ": obs"
        steps += 1
        if fx["done"][t]:
            break                          # the next episode starts from the reference's stream
    assert steps >= 1 and (name != "g4_respawn_in_conflict" or respawned.sum() == 1)
    if name == "g3_nmac_early_exit":       # intruders after the colliding one were not integrated (:167-168), the others were
        k = 3
        assert np.array_equal(st["int_x"][0][k + 1:].astype(np.float32), inj["int_x"][0][k + 1:].astype(np.float32))
        assert np.array_equal(st["int_x"][0][:k + 1].astype(np.float32),
                              (inj["int_x"][0][:k + 1].astype(np.float32) + inj["int_vx"][0][:k + 1]).astype(np.float32))
    if name == "g5_conflict_masks_goal":
        assert list(fx["info"][:steps]) == [1, 1, 1, 2] and bool(fx["done"][steps - 1])
    env.close()


def test_gym_style_env_is_a_drop_in(uamenv):
    """gym.Env duck type: signatures, types and info dict of Simulators/SingleAircraftEnv.py:64,126-139."""
    fx = gu.load("g1_discrete9_n4_seed0")
    env = uamenv.SingleAircraftEnv(num_intruders=4)
    assert env.seed(0) == [0]
    obs = env.reset()
    assert isinstance(obs, np.ndarray) and obs.dtype == np.float64 and obs.shape == (24,)
    assert env.observation_space.shape == (24,) and env.action_space.n == 9 and env.intruder_size == 4
    assert np.abs(obs - fx["obs0"]).max() <= 1e-5
    for t in range(64):
        obs, reward, done, info = env.step(int(fx["actions"][t]))
        assert isinstance(reward, float) and isinstance(done, bool) and set(info) == {"result"}
        assert np.abs(obs - fx["obs"][t]).max() <= 1e-5 and abs(reward - fx["reward"][t]) <= 1e-5
        assert info["result"] == ["", "c", "g", "n"][fx["info"][t]] and done == bool(fx["done"][t])
    assert env.no_conflict == int(fx["no_conflict"][63])
    assert env.render() is None
    env.close()
    ri = uamenv.SingleAircraftMCTSRandIntruderEnv(num_intruders=8)
    ri.seed(13)
    o = ri.reset()
    assert o.shape == (6 * 8 + 8,)
    o, r, d, info = ri.step((1, 2))
    assert isinstance(info, str)
    ri.close()


# ---------------------------------------------------------------- API behaviour
def test_packaged_2env_gym_class_is_a_drop_in(uamenv):
    """gym_guidance_collision_avoidance_single/envs/SingleAircraft2Env.py through the gym-style class: (a0, a1) action,
    bare-string info, its reward literals and the 'w' rule — the first episodes of the g8b fixture from seed(17)."""
    fx = gu.load("g8b_packaged_continuous2_n8_seed17")
    env = uamenv.PackagedSingleAircraft2Env(num_intruders=8)
    assert env.action_space.shape == (2,) and env.observation_space.shape == (40,)
    env.seed(17)
    assert np.abs(env.reset() - fx["obs0"]).max() <= 1e-5
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    seen = set()
    for t in range(400):
        ob, r, d, info = env.step(fx["actions"][t])
        assert isinstance(info, str) and info == ["", "c", "g", "n", "w"][int(fx["info"][t])], t
        assert d == bool(fx["done"][t]) and abs(r - fx["reward"][t]) <= 1e-5 and np.abs(ob - fx["obs"][t]).max() <= 1e-5, t
        seen.add(info)
        if d:
            assert np.abs(env.reset() - fx["reset_obs"][reset_steps[t]]).max() <= 1e-5
    assert "w" in seen, "the out-of-map termination was not exercised"
    with pytest.raises(AssertionError):
        env.step([2.0, 0.0])
    env.close()


def test_packaged_default_without_intruders(uamenv, oracle):
    """The packaged copy's own default, Config.intruder_size = 0 (envs/config.py:8): a map with the ownship and a goal
    only.  `PackagedSingleAircraft2Env()` — no arguments, as the reference's callers construct it — replays fixture g8c
    (recorded from the reference with N = 0: goals, out-of-map terminations, 8-wide observation) from seed(23); and the
    batched env without intruders matches the oracle in lockstep in both RNG modes, through resets, single steps and the
    looped kernel."""
    import torch
    fx = gu.load("g8c_packaged_default_n0_seed23")
    env = uamenv.PackagedSingleAircraft2Env()
    assert env.intruder_size == 0 and env.observation_space.shape == (8,)
    env.seed(23)
    assert np.abs(env.reset() - fx["obs0"]).max() <= 1e-5
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    seen = set()
    for t in range(len(fx["done"])):
        ob, r, d, info = env.step(fx["actions"][t])
        assert info == ["", "c", "g", "n", "w"][int(fx["info"][t])], t
        assert ob.shape == (8,) and d == bool(fx["done"][t]) and abs(r - fx["reward"][t]) <= 1e-5 and np.abs(ob - fx["obs"][t]).max() <= 1e-5, t
        seen.add(info)
        if d:
            assert np.abs(env.reset() - fx["reset_obs"][reset_steps[t]]).max() <= 1e-5
    assert {"g", "w"} <= seen
    env.close()
    for rng, precision in (("philox", "f32"), ("mt19937", "exact")):
        pu.run_lockstep(uamenv, oracle, 300, 0, 120, policy="goal", rng=rng, precision=precision, variant="continuous2",
                        wall_terminates=1, check_state_every=20)
    a = uamenv.BatchedSingleAircraftEnv(2000, 0, variant="discrete9", seed=4)
    b = uamenv.BatchedSingleAircraftEnv(2000, 0, variant="discrete9", seed=4)
    a.reset(); b.reset()
    tape = torch.stack([a.policy_actions("goal", seed=k) for k in range(30)]).contiguous()
    _, rew, done, info = a.step_many(tape)
    for k in range(30):
        o, r, d, i = b.step(tape[k])
        assert torch.equal(rew[k], r) and torch.equal(done[k], d.view(torch.uint8)), k
    assert torch.equal(a.observe(), b.observe())
    a.close(); b.close()


def test_reset_mask_state_roundtrip_and_step_many(uamenv, oracle):
    import torch
    B, N = 64, 8
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=5)
    env.reset()
    st0 = env.get_state()
    # masked reset touches only the masked envs
    mask = np.zeros(B, np.uint8)
    mask[::3] = 1
    before = env.observe().clone()
    after = env.reset(mask).clone()
    same = (before == after).all(dim=1).cpu().numpy()
    assert same[mask == 0].all() and not same[mask == 1].any()
    # set_state(get_state()) is the identity and the continuation is deterministic
    env.set_state(st0)
    tape = torch.randint(0, 9, (20, B), dtype=torch.int32, device=env.device)
    _, r1, d1, i1 = env.step_many(tape)
    o1 = env.observe().clone()
    env.set_state(st0)
    rs, ds = [], []
    for k in range(20):
        o, r, d, _ = env.step(tape[k])
        rs.append(r.clone())
        ds.append(d.clone())
    assert torch.equal(torch.stack(rs), r1) and torch.equal(torch.stack(ds).to(torch.uint8), d1)
    assert torch.equal(env.observe(), o1)
    env.close()


def test_bad_arguments_fail_loudly(uamenv):
    with pytest.raises(uamenv.UamError):
        uamenv.BatchedSingleAircraftEnv(8, 300)          # N > 256
    with pytest.raises(uamenv.UamError):
        uamenv.BatchedSingleAircraftEnv(0, 8)
    env = uamenv.BatchedSingleAircraftEnv(8, 8)
    with pytest.raises(ValueError):
        env.step(np.zeros(7, np.int32))
    with pytest.raises(uamenv.UamError):
        env.set_state({"int_promoted": np.ones((8, 8), np.uint8)})   # needs precision='exact'
    env.close()


# ---------------------------------------------------------------- full-size properties (BASELINE config 3)
def test_full_size_matches_threaded_oracle_and_is_shard_invariant(uamenv, oracle):
    """B=65536 x N=32 (the roofline configuration): a few steps against the 8-thread oracle, bit-exact
    checksums; then the same batch split over two handles by env_id_offset must give identical results
    (what the multi-GPU sharding relies on)."""
    import torch
    B, N, T = 65536, 32, 12
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=2024)
    ocfg = pu.oracle_cfg_from(oracle, env.cfg)
    orc = oracle.OracleBatch(ocfg, oracle.MATH_PORTABLE, threads=16)
    env.reset()
    orc.reset(want_obs=False)
    halves = [uamenv.BatchedSingleAircraftEnv(B // 2, N, seed=2024, env_id_offset=k * (B // 2)) for k in range(2)]
    for h in halves:
        h.reset()
    ost = orc.get_state()
    for t in range(T):
        acts = pu.goal_seek_actions(ost, env.cfg.d_heading, 0)
        o, r, d, inf = env.step(acts)
        oo = orc.step(acts, want_obs=False)
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), oo[2]) and np.array_equal(inf.codes.cpu().numpy(), oo[3])
        assert np.array_equal(r.cpu().numpy(), oo[1].astype(np.float32))
        ost = orc.get_state()
        for k, h in enumerate(halves):
            sl = slice(k * (B // 2), (k + 1) * (B // 2))
            ho, hr, hd, _ = h.step(acts[sl])
            assert torch.equal(ho, o[sl]) and torch.equal(hr, r[sl]) and torch.equal(hd, d[sl])
    pu.compare_states(env.get_state(), ost, "full size, after %d steps" % T)
    env.close()
    for h in halves:
        h.close()


def test_full_size_invariants_under_random_policy(uamenv):
    """Properties that need no oracle: determinism, bounds, flag/reward consistency, auto-reset."""
    import torch
    B, N, T = 65536, 32, 200
    runs = []
    for rep in range(2):
        env = uamenv.BatchedSingleAircraftEnv(B, N, seed=99)
        env.reset()
        acc = torch.zeros(4, dtype=torch.float64, device=env.device)
        for t in range(T):
            a = env.policy_actions("goal" if t % 2 else "random", seed=3)
            obs, rew, done, infos = env.step(a)
            codes = infos.codes
            # reward <-> info <-> done consistency (SingleAircraftEnv.py:167-185)
            assert bool(((codes == 3) == (rew == -1.0)).all()) and bool(((codes == 1) == (rew == -0.5)).all())
            assert bool((done == ((codes == 2) | (codes == 3))).all())
            assert bool((rew[codes == 0] <= 0).all()) and bool((rew[codes == 2] == 1.0).all())
            # normalised intruder positions stay inside the map (respawn on leaving it)
            pos = obs[:, :4 * N].reshape(B, N, 4)[:, :, :2]
            assert bool(((pos >= 0) & (pos <= 1)).all())
            # a done env was auto-reset: ownship back at (50,50)/800
            if bool(done.any()):
                assert bool((obs[done][:, 4 * N:4 * N + 2] == 0.0625).all())
            acc += torch.stack([rew.double().sum(), done.double().sum(), codes.double().sum(), obs.double().sum()])
        runs.append(acc.cpu().numpy())
        env.close()
    assert np.array_equal(runs[0], runs[1]), "two runs from the same seed differ: %s vs %s" % (runs[0], runs[1])
    assert runs[0][1] > 0, "no episode ended in %d steps" % T


# ---------------------------------------------------------------- edge shapes and fallback paths
@pytest.mark.parametrize("B,N,variant", [(1, 1, "discrete9"), (3, 3, "discrete9"), (1, 32, "discrete9"),
                                         (70, 5, "randintruder"),   # D = 6*5+8 = 38: rows not 16-B aligned -> scalar obs path
                                         (65, 7, "mcts"), (130, 33, "discrete9")])
def test_edge_shapes_match_oracle(uamenv, oracle, B, N, variant):
    pu.run_lockstep(uamenv, oracle, B, N, 150, policy="goal", variant=variant, rng="philox", precision="f32",
                    check_state_every=10)
    pu.run_lockstep(uamenv, oracle, B, N, 60, policy="goal", variant=variant, rng="mt19937", precision="exact",
                    check_state_every=10)


def test_unaligned_obs_pointer_uses_the_scalar_path(uamenv, oracle):
    """obs_out only 4-byte aligned: the library must fall back to per-element stores, same values."""
    import ctypes as C
    import torch
    B, N = 96, 8
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=11)
    ref = env.reset().clone()
    D = env.obs_dim
    raw = torch.zeros(B * D + 8, dtype=torch.float32, device=env.device)
    off = raw[1:1 + B * D]                      # +4 bytes: not 16-byte aligned
    assert off.data_ptr() % 16 != 0
    uamenv._capi.check(env._lib.uam_observe(env._h, C.c_void_p(off.data_ptr()), env._stream()))
    assert torch.equal(off.view(B, D), ref)
    acts = torch.randint(0, 9, (B,), dtype=torch.int32, device=env.device)
    st = env.get_state()
    o1 = env.step(acts)[0].clone()
    env.set_state(st)
    uamenv._capi.check(env._lib.uam_step(env._h, C.c_void_p(acts.data_ptr()), C.c_void_p(off.data_ptr()),
                                         C.c_void_p(env._rew.data_ptr()), C.c_void_p(env._done.data_ptr()),
                                         C.c_void_p(env._info.data_ptr()), env._stream()))
    assert torch.equal(off.view(B, D), o1)
    assert float(raw[0]) == 0.0 and float(raw[1 + B * D]) == 0.0      # no write outside the buffer
    env.close()


def test_mt19937_state_exports_in_numpy_layout(uamenv, oracle):
    """get_state() finishes the lazily twisted round: key/pos/gauss equal the oracle's NumPy-format state,
    and importing a NumPy-format state continues the same stream."""
    B, N = 16, 8
    env = uamenv.BatchedSingleAircraftEnv(B, N, rng="mt19937", precision="exact", seed=321, auto_reset=True)
    orc = oracle.OracleBatch(pu.oracle_cfg_from(oracle, env.cfg), oracle.MATH_PORTABLE)
    env.reset()
    orc.reset()
    rs = np.random.RandomState(0)
    for t in range(140):                          # crosses a 624-word round (about 5 words per step + spawns)
        a = rs.randint(0, 9, size=B).astype(np.int32)
        env.step(a)
        orc.step(a)
    gs, os_ = env.get_state(), orc.get_state()
    for k in ("mt_key", "mt_pos", "mt_has_gauss", "mt_gauss"):
        assert np.array_equal(gs[k], os_[k]), k
    # round trip through the NumPy layout, then keep stepping in lockstep
    env.set_state(gs)
    for t in range(40):
        a = rs.randint(0, 9, size=B).astype(np.int32)
        o, r, d, inf = env.step(a)
        oo = orc.step(a)
        assert np.array_equal(r.cpu().numpy(), oo[1].astype(np.float32)) and np.array_equal(d.cpu().numpy().astype(np.uint8), oo[2])
    pu.compare_states(env.get_state(), orc.get_state(), "after numpy-layout round trip")
    env.close()


def test_seed_array_and_reseed(uamenv, oracle):
    """Per-env seeds: env e of a batch equals a single env seeded the same (MT19937 mode)."""
    seeds = np.array([5, 99, 5, 12345], dtype=np.uint64)
    env = uamenv.BatchedSingleAircraftEnv(4, 8, rng="mt19937", precision="exact", auto_reset=False)
    env.seed(seeds)
    obs = env.reset().cpu().numpy()
    assert np.array_equal(obs[0], obs[2]) and not np.array_equal(obs[0], obs[1])
    fx = gu.load("g2_goalseek_n8_seed3")
    env.seed(np.array([3, 3, 3, 3], dtype=np.uint64))
    obs = env.reset().cpu().numpy()
    assert np.abs(obs[1] - fx["obs0"]).max() <= 1e-5
    with pytest.raises(ValueError):          # np.random.seed raises beyond 32 bits; silently truncating would alias seeds
        env.seed([1 << 32, 1, 2, 3])
    with pytest.raises(ValueError):
        env.seed([-1, 1, 2, 3])
    env.close()
    ph = uamenv.BatchedSingleAircraftEnv(4, 8)
    with pytest.raises(ValueError):          # one Philox key per batch: a per-env seed array would be silently ignored
        ph.seed([1, 2, 3, 4])
    ph.seed(9); a = ph.reset().clone()
    ph.seed([9, 9, 9, 9]); b = ph.reset().clone()
    ph.seed(10); c = ph.reset().clone()
    import torch
    assert torch.equal(a, b) and not torch.equal(a, c)
    ph.close()


def test_monitor_episode_stats_match_oracle(uamenv, oracle):
    """Monitor-style accumulators (baselines/bench/monitor.py:58-75): return and length of the episode that
    just ended, emitted on the done step, against the oracle's; and the lazily built info['episode']."""
    B, N, T = 256, 8, 400
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=77, monitor=True)
    orc = oracle.OracleBatch(pu.oracle_cfg_from(oracle, env.cfg), oracle.MATH_PORTABLE)
    env.reset()
    orc.reset()
    ost = orc.get_state()
    finished = 0
    acc_r, acc_l = np.zeros(B), np.zeros(B, dtype=np.int64)
    for t in range(T):
        acts = pu.goal_seek_actions(ost, env.cfg.d_heading, 0)
        obs, rew, done, infos = env.step(acts)
        oo = orc.step(acts)
        ost = orc.get_state()
        d = done.cpu().numpy()
        acc_r += rew.cpu().numpy().astype(np.float64)
        acc_l += 1
        if d.any():
            gr, gl = env.episode_stats()
            gr, gl = gr.cpu().numpy(), gl.cpu().numpy()
            orr, ol = orc.episode_stats()
            assert np.array_equal(gl[d], ol[d]) and np.array_equal(gr[d], orr[d])       # same float32 rewards summed in order
            assert np.array_equal(gl[d], acc_l[d]) and np.allclose(gr[d], acc_r[d], rtol=0, atol=1e-9)
            i = int(np.nonzero(d)[0][0])
            assert infos[i]["episode"] == {"r": round(float(gr[i]), 6), "l": int(gl[i])} and "episode" not in infos[int(np.nonzero(~d)[0][0])]
            finished += int(d.sum())
            acc_r[d] = 0
            acc_l[d] = 0
    assert finished > 50
    plain = uamenv.BatchedSingleAircraftEnv(4, 4)
    with pytest.raises(uamenv.UamError):
        plain.episode_stats()
    plain.close()
    env.close()


# ---------------------------------------------------------------- k-nearest observation / HER env
@pytest.mark.parametrize("rng,precision,N", [("philox", "f32", 8), ("philox", "exact", 32), ("mt19937", "exact", 8),
                                              ("philox", "f32", 80)])
def test_her9_knn_obs_matches_oracle(uamenv, oracle, rng, precision, N):
    pu.run_lockstep(uamenv, oracle, 160, N, 250, policy="goal", variant="her9", rng=rng, precision=precision,
                    check_state_every=10)


def test_her9_wall_termination_matches_oracle(uamenv, oracle):
    class Cfg(uamenv.Config):
        sparse_reward = True
    out = pu.run_lockstep(uamenv, oracle, 256, 8, 400, policy="random", variant="her9", rng="philox", precision="f32",
                          config=Cfg, check_state_every=10)
    assert out["counts"].shape[0] >= 5 and out["counts"][4] > 0, "no wall termination exercised: %s" % out["counts"]


@pytest.mark.parametrize("name", [n for n in gu.HER_FIXTURES if n.startswith("g13_")])
def test_gpu_replays_her_fixture_from_seed(uamenv, name):
    """The HER env's reference trajectories (k-nearest ordering, 'w' terminations) from np.random.seed(s) alone."""
    fx = gu.load(name)
    N = int(fx["meta_N"])
    over = gu.config_overrides(fx)
    env = uamenv.BatchedSingleAircraftEnv(1, N, variant="her9", rng="mt19937", precision="exact", seed=int(fx["meta_seed"]),
                                          auto_reset=False, sparse_reward=over["sparse_reward"],
                                          wall_terminates=over["wall_terminates"], knn=over["knn"])
    env.reset()                       # the reference's constructor resets once
    tol = 1e-5
    obs0 = env.reset().cpu().numpy()[0]
    assert np.abs(obs0 - fx["obs0"]).max() <= tol
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    for t in range(len(fx["actions"])):
        obs, rew, done, infos = env.step(fx["actions"][t:t + 1])
        where = "%s step %d" % (name, t)
        assert bool(done[0].item()) == bool(fx["done"][t]) and int(infos.codes[0].item()) == int(fx["info"][t]), where
        assert abs(float(rew[0].item()) - fx["reward"][t]) <= tol, where
        assert np.abs(obs.cpu().numpy()[0] - fx["obs"][t]).max() <= tol, where + " obs"
        if fx["done"][t]:
            assert np.abs(env.reset().cpu().numpy()[0] - fx["reset_obs"][reset_steps[t]]).max() <= tol, where + " reset obs"
    env.close()


def test_her9_gym_class(uamenv):
    fx = gu.load("g13_her9_n8_seed41")
    env = uamenv.SingleAircraftDiscrete9HEREnv(num_intruders=8)
    env.seed(41)
    ob = env.reset()
    assert set(ob) == {"observation", "achieved_goal", "desired_goal"} and ob["observation"].shape == (24,)
    assert env.observation_space.shape == (26,)
    flat = np.concatenate([ob["observation"], ob["achieved_goal"], ob["desired_goal"]])
    assert np.abs(flat - fx["obs0"]).max() <= 1e-5
    ob, r, d, info = env.step(int(fx["actions"][0]))
    assert abs(r - fx["reward"][0]) <= 1e-5 and info == {"result": ""}
    assert isinstance(env.compute_reward(ob["achieved_goal"], ob["desired_goal"], None), float)
    env.close()


# ---------------------------------------------------------------- Discrete3HER env: 3 actions, shaped reward, goal margin
@pytest.mark.parametrize("rng,precision,N,sparse", [("philox", "f32", 8, False), ("philox", "f32", 32, True),
                                                     ("philox", "exact", 8, False), ("mt19937", "exact", 8, True),
                                                     ("mt19937", "f32", 80, False)])
def test_her3_shaped_reward_matches_oracle(uamenv, oracle, rng, precision, N, sparse):
    """Sparse: the float32 shaping is the reference's own arithmetic -> rewards bit-exact; dense: the GPU adds the
    shaping to the float32 dense term instead of rounding the float64 sum once -> <= 1 float32 ulp (tolerance 1e-6)."""
    class Cfg(uamenv.Config):
        sparse_reward = sparse
    out = pu.run_lockstep(uamenv, oracle, 192, N, 300, policy="mixed", variant="her3", rng=rng, precision=precision,
                          config=Cfg, check_state_every=10, reward_tol=0.0 if sparse else 1e-6)
    assert out["counts"][1] > 0 and out["counts"][3] > 0, out["counts"]


@pytest.mark.parametrize("name", [n for n in gu.HER_FIXTURES if n.startswith("g14_")])
def test_gpu_replays_her3_fixture_from_seed(uamenv, name):
    """Discrete3HER reference trajectories from np.random.seed(s) alone: obs, shaped rewards, codes, nearest distance."""
    fx = gu.load(name)
    N = int(fx["meta_N"])
    over = gu.config_overrides(fx)
    env = uamenv.BatchedSingleAircraftEnv(1, N, variant="her3", rng="mt19937", precision="exact", seed=int(fx["meta_seed"]),
                                          auto_reset=False, sparse_reward=over["sparse_reward"],
                                          wall_terminates=over["wall_terminates"], knn=over["knn"])
    assert env.cfg.shaped_reward == 1 and env.cfg.goal_margin == 100.0 and env.cfg.conflict_coeff == over["conflict_coeff"]
    env.reset()                       # the reference's constructor resets once
    tol = 1e-5
    assert np.abs(env.reset().cpu().numpy()[0] - fx["obs0"]).max() <= tol
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    for t in range(len(fx["actions"])):
        obs, rew, done, infos = env.step(fx["actions"][t:t + 1])
        where = "%s step %d" % (name, t)
        assert bool(done[0].item()) == bool(fx["done"][t]) and int(infos.codes[0].item()) == int(fx["info"][t]), where
        assert abs(float(rew[0].item()) - fx["reward"][t]) <= tol, where
        assert abs(float(env.nearest()[0].item()) - fx["nearest"][t]) <= tol * fx["nearest"][t], where + " nearest"
        assert np.abs(obs.cpu().numpy()[0] - fx["obs"][t]).max() <= tol, where + " obs"
        if fx["done"][t]:
            assert np.abs(env.reset().cpu().numpy()[0] - fx["reset_obs"][reset_steps[t]]).max() <= tol, where + " reset obs"
    env.close()


def test_her3_gym_class(uamenv):
    fx = gu.load("g14_her3_n8_seed51")
    env = uamenv.SingleAircraftDiscrete3HEREnv(num_intruders=8)
    env.seed(51)
    ob = env.reset()
    assert env.action_space.n == 3 and env.observation_space.shape == (26,)
    flat = np.concatenate([ob["observation"], ob["achieved_goal"], ob["desired_goal"]])
    assert np.abs(flat - fx["obs0"]).max() <= 1e-5
    ob, r, d, near = env.step(int(fx["actions"][0]))
    assert abs(r - fx["reward"][0]) <= 1e-5 and isinstance(near, float) and abs(near - fx["nearest"][0]) <= 1e-3
    env.close()


@pytest.mark.parametrize("sparse", [False, True])
def test_her_relabel_matches_reference_and_oracle(uamenv, oracle, sparse):
    """uam_her_goal_reward / uam_her_input_reward against the reference's own outputs (g15) and the oracle."""
    import torch
    fx = gu.load("g15_her_relabel")
    X = torch.from_numpy(fx["inputs"]).cuda()
    tag = "sparse" if sparse else "dense"
    rew, done = uamenv.her_input_reward(X, sparse_reward=sparse, num_intruders=int(fx["meta_N"]), knn=int(fx["meta_knn"]))
    want = fx["input_reward9_" + tag]
    assert np.abs(rew.cpu().numpy() - want).max() <= 1e-6
    assert np.array_equal(done.cpu().numpy(), (want == 1.0) | (want == -1.0))       # agent_her.py:117 (r_n == 10 or -10, scaled)
    cat = lambda r: np.select([r == 1.0, r == -1.0, r == -0.5], [1, 2, 3], 0)
    assert np.array_equal(cat(rew.cpu().numpy().astype(np.float64)), cat(want)), "reward categories must be exact"
    g = uamenv.her_goal_reward(X[:, :2], X[:, -2:], sparse_reward=sparse)
    assert np.abs(g.cpu().numpy() - fx["goal_reward9_" + tag]).max() <= 1e-6
    # a larger random batch against the oracle (portable math = the device's x * x convention): bit-exact after the float32 cast
    rs = np.random.RandomState(3)
    big = rs.uniform(0, 1, size=(20000, 26)).astype(np.float32)
    big[::3, 4:6] = big[::3, 0:2] + rs.uniform(-0.03, 0.03, size=(len(big[::3]), 2)).astype(np.float32)
    big[1::3, -2:] = big[1::3, 0:2] + rs.uniform(-0.03, 0.03, size=(len(big[1::3]), 2)).astype(np.float32)
    ocfg = oracle.default_config(1, 8, obs_mode=oracle.OBS_KNN5, knn=4, sparse_reward=int(sparse))
    rew, _ = uamenv.her_input_reward(torch.from_numpy(big).cuda(), sparse_reward=sparse, num_intruders=8, knn=4)
    want = oracle.her_input_reward(ocfg, big.astype(np.float64), oracle.MATH_PORTABLE).astype(np.float32)
    assert np.array_equal(rew.cpu().numpy(), want)
    assert len(set(np.round(want[want > -0.6], 3).tolist()) & {1.0, -0.5}) == 2 and (want == -1.0).any()
    flat = torch.empty(big.size + 1, dtype=torch.float32, device="cuda")       # 4-byte-aligned rows: the un-staged kernel
    flat[1:] = torch.from_numpy(big).cuda().reshape(-1)
    rew2, _ = uamenv.her_input_reward(flat[1:].view(big.shape), sparse_reward=sparse, num_intruders=8, knn=4)
    assert np.array_equal(rew2.cpu().numpy(), want)
    g = uamenv.her_goal_reward(torch.from_numpy(big[:, :2].copy()).cuda(), torch.from_numpy(big[:, -2:].copy()).cuda(), sparse_reward=sparse)
    assert np.array_equal(g.cpu().numpy(), oracle.her_goal_reward(ocfg, big[:, :2], big[:, -2:]).astype(np.float32))
    # the two-rows-per-thread kernel's odd last row, and 8-byte-aligned inputs (one row per thread)
    odd = big[:19999]
    want_g = oracle.her_goal_reward(ocfg, odd[:, :2], odd[:, -2:]).astype(np.float32)
    g = uamenv.her_goal_reward(torch.from_numpy(odd[:, :2].copy()).cuda(), torch.from_numpy(odd[:, -2:].copy()).cuda(), sparse_reward=sparse)
    assert np.array_equal(g.cpu().numpy(), want_g)
    pad_a = torch.empty(2 * 19999 + 2, dtype=torch.float32, device="cuda"); pad_g = torch.empty_like(pad_a)
    pad_a[2:] = torch.from_numpy(odd[:, :2].copy()).cuda().reshape(-1); pad_g[2:] = torch.from_numpy(odd[:, -2:].copy()).cuda().reshape(-1)
    g = uamenv.her_goal_reward(pad_a[2:].view(-1, 2), pad_g[2:].view(-1, 2), sparse_reward=sparse)
    assert np.array_equal(g.cpu().numpy(), want_g)


# ---------------------------------------------------------------- NumPy-facing VecEnv adapter
def test_numpy_vecenv_adapter_matches_device_path_and_monitor(uamenv):
    """UamVecEnv = the baselines VecEnv contract with host arrays: same numbers as the device-tensor path on the same
    seed and actions; dtypes as DummyVecEnv produces them; info['episode'] consistent with the rewards observed."""
    B, N, T = 96, 8, 260
    venv = uamenv.UamVecEnv(B, N, seed=5, monitor=True)
    ref = uamenv.BatchedSingleAircraftEnv(B, N, seed=5, monitor=True)
    assert venv.num_envs == B and venv.observation_space.shape == (4 * N + 8,) and venv.action_space.n == 9
    o = venv.reset()
    assert isinstance(o, np.ndarray) and o.dtype == np.float32 and o.shape == (B, 4 * N + 8)
    assert np.array_equal(o, ref.reset().cpu().numpy())
    rs = np.random.RandomState(0)
    st = ref.get_state()
    ret, ln, eps = np.zeros(B), np.zeros(B, dtype=np.int64), 0
    for t in range(T):
        acts = pu.goal_seek_actions(st, ref.cfg.d_heading, 0)
        acts[rs.uniform(size=B) < 0.1] = rs.randint(0, 9)
        venv.step_async(acts.astype(np.int64))                      # any integer dtype, like a Python learner would pass
        obs, rew, done, infos = venv.step_wait()
        o2, r2, d2, i2 = ref.step(acts)
        assert obs.dtype == np.float32 and rew.dtype == np.float32 and done.dtype == np.bool_ and len(infos) == B
        assert np.array_equal(obs, o2.cpu().numpy()) and np.array_equal(rew, r2.cpu().numpy())
        assert np.array_equal(done, d2.cpu().numpy()) and np.array_equal(infos.codes, i2.codes.cpu().numpy())
        ret += rew.astype(np.float64); ln += 1
        for i, r, l in infos.episodes():
            assert done[i] and abs(r - ret[i]) < 1e-9 and l == ln[i]
            assert infos[i]["episode"]["l"] == l and infos[i]["result"] in ("g", "n")
            eps += 1
        ret[done] = 0.0; ln[done] = 0
        st = ref.get_state()
    assert eps > 0 and infos[0].keys() >= {"result"}
    assert venv.get_images() == [] and venv.unwrapped is venv
    venv.close(); ref.close()


# ---------------------------------------------------------------- uam_step_many: K steps in one launch
@pytest.mark.parametrize("variant,B,N,K", [("discrete9", 4096, 8, 7), ("discrete9", 300, 32, 16), ("randintruder", 512, 64, 5),
                                           ("her9", 700, 20, 9), ("her3", 257, 8, 12), ("random_start", 1000, 4, 33),
                                           ("continuous", 640, 32, 3), ("discrete9", 65536, 32, 4)])
def test_step_many_single_launch_equals_single_steps(uamenv, variant, B, N, K):
    """The looped kernel (small grids) and the loop of launches (large grids, last case) must both reproduce K
    separate uam_step calls bit for bit: rewards, dones, infos, the final observation and the whole state."""
    import torch
    a = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=11, monitor=True)
    b = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=11, monitor=True)
    a.reset(); b.reset()
    for rep in range(3):
        tape = torch.stack([a.policy_actions("goal" if rep else "random", seed=rep * 100 + k) for k in range(K)])
        obs_m, rew_m, done_m, info_m = a.step_many(tape)
        for k in range(K):
            o, r, d, i = b.step(tape[k])
            assert torch.equal(rew_m[k], r) and torch.equal(done_m[k], d.view(torch.uint8)) and torch.equal(info_m[k], i.codes), (rep, k)
        assert torch.equal(obs_m, o)
        pu.compare_states(a.get_state(), b.get_state(), "after step_many rep %d" % rep,
                          hold_sh=variant == "randintruder")
        ra, la = a.episode_stats(); rb, lb = b.episode_stats()
        assert torch.equal(ra, rb) and torch.equal(la, lb)
    if variant == "her3":
        assert torch.equal(a.nearest(), b.nearest())
    a.close(); b.close()


# ---------------------------------------------------------------- BASELINE configs 4 and 5 at their stated shapes
def test_config4_continuous_65536x32_sharded_8_ways_matches_oracle(uamenv, oracle):
    """BASELINE.json configs[3]: SingleAircraft2Env continuous, ONE batch of 65536 x 32 partitioned 8192 per GPU.  On
    the one-GPU box the eight shards are eight handles (env_id_offset = first global env), each stepping its chunk
    the way a rank of the 8-GPU run does (uam_rollout: one looped launch per chunk); their union, and one unsharded
    65536-env handle stepped launch by launch, must both equal the threaded oracle over the whole batch: flags and
    rewards bit-exact, observations within 1e-6, the whole float state bit-exact."""
    import torch
    from uamenv.sharded import shard_bounds
    B, N, K, W = 65536, 32, 10, 8
    whole = uamenv.BatchedSingleAircraftEnv(B, N, variant="continuous", seed=404, fuse_rollout="never")    # launch by launch
    ocfg = pu.oracle_cfg_from(oracle, whole.cfg)
    orc = oracle.OracleBatch(ocfg, oracle.MATH_PORTABLE, threads=16)
    spans = [shard_bounds(B, W, r) for r in range(W)]
    shards = [uamenv.BatchedSingleAircraftEnv(c, N, variant="continuous", seed=404, env_id_offset=f) for f, c in spans]
    assert all(c == 8192 for _, c in spans) and all(s.steps_per_launch(K) == K for s in shards)
    assert whole.steps_per_launch(K) == 1
    o0 = whole.reset().cpu().numpy()
    oo0 = orc.reset()
    assert np.abs(o0 - oo0).max() <= 1e-6
    for s, (f, c) in zip(shards, spans):
        assert torch.equal(s.reset(), whole._obs[f:f + c])
    rs = np.random.RandomState(12)
    ost = orc.get_state()
    # fly the batch into mid-episode first (conflicts, goals, NMACs, respawns all occur), unsharded vs oracle; the
    # shards then take over the state of their slice — exactly what a checkpointed sharded run would do
    for t in range(160):
        a = pu.goal_seek_actions(ost, whole.cfg.d_heading, 2)
        o, r, d, inf = whole.step(a)
        oo = orc.step(a, want_obs=False)
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), oo[2]) and np.array_equal(inf.codes.cpu().numpy(), oo[3]), t
        ost = orc.get_state()
    wst = whole.get_state()
    pu.compare_states(wst, ost, "config 4, unsharded, after the warm-up")
    for s, (f, c) in zip(shards, spans):
        s.set_state({k: (v if k == "step_index" else np.ascontiguousarray(v[f:f + c])) for k, v in wst.items()})
    tape = np.zeros((K, B), dtype=np.float32)
    orew, odone, oinfo = [], [], []
    for k in range(K):
        a = pu.goal_seek_actions(ost, whole.cfg.d_heading, 2)
        m = rs.uniform(size=B) < 0.3
        a[m] = rs.uniform(-1, 1, size=int(m.sum())).astype(np.float32)
        tape[k] = a
        o, r, d, inf = whole.step(a)
        oo = orc.step(a)
        pu.compare_outputs((o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), inf.codes.cpu().numpy()), oo, "config 4 step %d" % k, 1e-6)
        orew.append(oo[1].astype(np.float32)); odone.append(oo[2]); oinfo.append(oo[3])
        ost = orc.get_state()
    pu.compare_states(whole.get_state(), ost, "config 4, unsharded, after %d steps" % K)
    orew, odone, oinfo = np.stack(orew), np.stack(odone), np.stack(oinfo)
    assert (oinfo == 1).any() and (oinfo == 2).any() and odone.any(), "the chunk should see conflicts, goals and resets"
    tape_d = torch.from_numpy(tape).to(whole.device)
    for s, (f, c) in zip(shards, spans):
        rew = torch.empty((K, c), dtype=torch.float32, device=s.device)
        done = torch.empty((K, c), dtype=torch.uint8, device=s.device)
        info = torch.empty((K, c), dtype=torch.uint8, device=s.device)
        obs = s.rollout_into(tape_d[:, f:f + c].contiguous(), rew, done, info)
        assert np.array_equal(rew.cpu().numpy(), orew[:, f:f + c]) and np.array_equal(done.cpu().numpy(), odone[:, f:f + c])
        assert np.array_equal(info.cpu().numpy(), oinfo[:, f:f + c])
        assert torch.equal(obs[0], whole._obs[f:f + c])               # the chunk's last observation
        sst = s.get_state()
        for key in pu.STATE_EXACT:
            if key in ("step_index",):
                continue
            assert np.array_equal(sst[key], np.asarray(ost[key])[f:f + c].astype(sst[key].dtype)), (f, key)
        s.close()
    whole.close(); orc.close()


def test_config5_randintruder_8192x64_steps_and_one_mcts_plan_match_oracle(uamenv, oracle):
    """BASELINE.json configs[4]: SingleAircraftMCTSRandIntruderEnv, 8192 envs x 64 intruders, driven by the MCTS
    planner (Algorithms/MCTS/Agent_RandInt.py:25-46).  Five lockstep steps against the oracle at the full shape; the
    first action of every env comes from ONE batched best_action(100, 3) over all 8192 roots, checked against the
    oracle's search on a sample of roots (child visit counts, q and the chosen action bit-exact)."""
    import torch
    B, N = 8192, 64
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant="randintruder", seed=77)
    ocfg = pu.oracle_cfg_from(oracle, env.cfg)
    orc = oracle.OracleBatch(ocfg, oracle.MATH_PORTABLE, threads=16)
    go, oo = env.reset().cpu().numpy(), orc.reset()
    assert (np.abs(go - oo) <= 1e-6 * np.maximum(1.0, np.abs(oo))).all()
    planner = uamenv.BatchedMCTS(N, seed=7, simulations=100, search_depth=3)
    roots = planner.roots_from_env(env)
    acts, cq, cn = planner.best_action(roots, root_id0=0, want_children=True)
    torch.cuda.synchronize()
    roots_h, acts_h, cq_h, cn_h = roots.cpu().numpy(), acts.cpu().numpy(), cq.cpu().numpy(), cn.cpu().numpy()
    assert acts_h.shape == (B, 2) and ((acts_h >= 0) & (acts_h <= 2)).all()
    mo = oracle.UamMctsConfig()
    for name, _ in oracle.UamMctsConfig._fields_:
        setattr(mo, name, getattr(planner.cfg, name))
    for k in (0, 1, 511, 4095, 4096, 8191) + tuple(np.random.RandomState(3).randint(0, B, 10)):
        o = oracle.mcts_best_action(mo, roots_h[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=int(k))
        assert np.array_equal(cn_h[k], o["child_n"]) and np.array_equal(cq_h[k], o["child_q"]), k
        assert np.array_equal(acts_h[k], o["best_action"]), k
    rs = np.random.RandomState(8)
    a = acts_h.astype(np.int32)
    for t in range(5):
        o, r, d, inf = env.step(a)
        oo = orc.step(a)
        pu.compare_outputs((o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), inf.codes.cpu().numpy()), oo,
                           "config 5 step %d" % t, 1e-6)
        a = pu.random_actions(rs, B, 1)
    pu.compare_states(env.get_state(), orc.get_state(), "config 5 after 5 steps", hold_sh=True)
    env.close(); orc.close()


# ---------------------------------------------------------------- G10: B envs on ONE NumPy stream, DummyVecEnv order
def test_shared_stream_batch_replays_the_reference_process(uamenv, oracle):
    """`rng="mt19937_shared"`: what four env objects in ONE reference process do to the global np.random after
    np.random.seed(5), stepped round-robin with reset-on-done inside each env's turn (dummy_vec_env.py:47-55) —
    "identical RNG seed" for a batch.  The fixture g10 was recorded that way from the reference."""
    fx = gu.load("g10_shared_stream_b4_n8_seed5")
    B, N = int(fx["meta_B"]), int(fx["meta_N"])
    env = uamenv.BatchedSingleAircraftEnv(B, N, rng="mt19937_shared", precision="exact", seed=int(fx["meta_seed"]), auto_reset=True)
    obs0 = env.reset().cpu().numpy()
    assert np.abs(obs0 - fx["obs0"]).max() <= 1e-5
    for t in range(fx["actions"].shape[0]):
        obs, rew, done, infos = env.step(fx["actions"][t])
        where = "g10 step %d" % t
        assert np.array_equal(done.cpu().numpy().astype(np.uint8), fx["done"][t]), where
        assert np.array_equal(infos.codes.cpu().numpy(), fx["info"][t]), where
        assert np.abs(rew.cpu().numpy() - fx["reward"][t]).max() <= 1e-5, where
        assert np.abs(obs.cpu().numpy() - fx["obs_after"][t]).max() <= 1e-5, where
        nc = env.no_conflict.cpu().numpy()
        assert np.array_equal(np.where(fx["done"][t] != 0, 0, nc), np.where(fx["done"][t] != 0, 0, fx["no_conflict"][t])), where
    assert fx["done"].sum() > 0
    env.close()
    # and against the oracle in the same mode, bit for bit, at a larger batch and through the RandIntruder draws
    pu.run_lockstep(uamenv, oracle, 24, 8, 120, policy="goal", rng="mt19937_shared", precision="exact", check_state_every=10)
    pu.run_lockstep(uamenv, oracle, 6, 20, 60, policy="goal", rng="mt19937_shared", precision="exact", variant="randintruder",
                    check_state_every=10)
    with pytest.raises(uamenv.UamError):
        uamenv.BatchedSingleAircraftEnv(2048, 8, rng="mt19937_shared", precision="exact")


def test_benchmarked_instance_spawn_law_matches_the_reference_stream(uamenv):
    """The benchmarked instance (Philox, float32, the lean spawn: 24-bit uniforms, float32 quadrant sincos) is
    builder-defined on both sides of its lockstep tests, so here its SPAWN LAW is held against the reference's:
    analytically (random_pos / random_speed / random_heading and the redraw-while-too-close rule,
    Simulators/SingleAircraftEnv.py:230-251, 270-279) and empirically, two-sample, against the MT19937 / exact
    instance, which replays the reference's own stream seed for seed."""
    from scipy import stats
    B, N = 8192, 32
    cfg = uamenv.Config
    W, H = float(cfg.window_width), float(cfg.window_height)

    def spawn_sample(**kw):
        env = uamenv.BatchedSingleAircraftEnv(B, N, variant="randintruder", seed=123, **kw)
        env.reset()
        st = env.get_state(["int_x", "int_y", "int_vx", "int_vy", "int_speed", "int_heading", "own_x", "own_y"])
        env.close()
        return {k: np.asarray(v, dtype=np.float64) for k, v in st.items() if k != "step_index"}

    lean = spawn_sample(rng="philox", precision="f32")
    ref = spawn_sample(rng="mt19937", precision="exact")
    for s in (lean, ref):
        x, y = s["int_x"].ravel(), s["int_y"].ravel()
        ox, oy = np.repeat(s["own_x"], N), np.repeat(s["own_y"], N)
        # inside the map, never closer to the ownship than initial_min_dist
        assert x.min() >= 0 and x.max() <= W and y.min() >= 0 and y.max() <= H
        assert np.hypot(x - ox, y - oy).min() >= cfg.initial_min_dist - 1e-3
        # speed and heading uniform on their ranges; the velocity is the held speed along the held heading
        sp, hd = s["int_speed"].ravel(), s["int_heading"].ravel()
        assert stats.kstest((sp - cfg.min_speed) / (cfg.max_speed - cfg.min_speed), "uniform").pvalue > 1e-3
        assert stats.kstest(hd / (2 * np.pi), "uniform").pvalue > 1e-3
        vx, vy = s["int_vx"].ravel(), s["int_vy"].ravel()
        assert np.abs(vx - sp * np.cos(hd)).max() < 2e-6 and np.abs(vy - sp * np.sin(hd)).max() < 2e-6
        # positions uniform on the map outside the exclusion disc: away from the disc's bounding box both marginals are
        far = (x > ox + cfg.initial_min_dist) & (y > oy + cfg.initial_min_dist)
        lo_x, lo_y = (ox + cfg.initial_min_dist)[far], (oy + cfg.initial_min_dist)[far]
        assert stats.kstest((x[far] - lo_x) / (W - lo_x), "uniform").pvalue > 1e-3
        assert stats.kstest((y[far] - lo_y) / (H - lo_y), "uniform").pvalue > 1e-3
    # and the two instances draw from the same law
    for k in ("int_x", "int_y", "int_speed", "int_heading", "int_vx", "int_vy"):
        assert stats.ks_2samp(lean[k].ravel(), ref[k].ravel()).pvalue > 1e-3, k
    # the fraction of spawns that needed a redraw is the disc's share of the map in both
    # (not observable from the state; the promoted-position flag of the exact instance is its trace)


def test_benchmarked_instance_heading_change_law_matches_the_reference_stream(uamenv):
    """RandIntruder's _update_headings (SingleAircraftMCTSRandIntruderEnv.py:166-174, 332-336) on the benchmarked
    instance — one Philox block per quad, a change's second uniform taken as u / p — against its law (each intruder
    changes with probability heading_change_prob per step, by a uniform angle within +-heading_change_max_deg) and,
    two-sample, against the MT19937 / exact instance that replays the reference's stream."""
    from scipy import stats
    import torch
    B, N, T = 2048, 32, 12
    prob, max_deg = 0.1, 10.0          # Simulators/config.py: the RandIntruder env's constants (uam_config defaults)

    def deltas(**kw):
        env = uamenv.BatchedSingleAircraftEnv(B, N, variant="randintruder", seed=77, **kw)
        assert abs(env.cfg.heading_change_prob - prob) < 1e-12 and abs(env.cfg.heading_change_max_deg - max_deg) < 1e-12
        env.reset()
        fields = ["int_x", "int_y", "int_heading"]
        prev = env.get_state(fields)
        changed, total, dd = 0, 0, []
        for t in range(T):
            env.step(env.policy_actions("goal", seed=t + 1))
            cur = env.get_state(fields)
            # intruders that were neither respawned nor reset in this step moved by their velocity (< 3 per step)
            cont = np.hypot(cur["int_x"].astype(np.float64) - prev["int_x"], cur["int_y"].astype(np.float64) - prev["int_y"]) < 4.0
            dh = (cur["int_heading"] - prev["int_heading"])[cont]
            total += dh.size
            changed += int((dh != 0).sum())
            dd.append(dh[dh != 0])
            prev = cur
        env.close()
        return changed, total, np.degrees(np.concatenate(dd))

    out = {}
    for name, kw in (("lean", dict(rng="philox", precision="f32")), ("ref", dict(rng="mt19937", precision="exact"))):
        changed, total, dd = deltas(**kw)
        rate = changed / total
        assert abs(rate - prob) < 5 * np.sqrt(prob * (1 - prob) / total), (name, rate)
        assert np.abs(dd).max() <= max_deg + 1e-9
        assert stats.kstest((dd + max_deg) / (2 * max_deg), "uniform").pvalue > 1e-3, name
        out[name] = dd
    assert stats.ks_2samp(out["lean"], out["ref"]).pvalue > 1e-3
