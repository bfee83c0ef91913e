"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz).

Every fixture is replayed from the seed alone (NumPy-legacy MT19937 stream, libm
trig) and must match the reference BIT-EXACTLY: float64 observations, rewards,
flags, counters and the full state including the float32/float64 position
promotion quirk.  This is what makes the oracle a trustworthy checker for the
HIP path (the reference has no tests of its own for this path).
"""
import numpy as np
import pytest

import golden_util as gu


def _assert_state(orc_state, want, where):
    for k, v in want.items():
        got = orc_state[k]
        assert got.shape == v.shape, (where, k, got.shape, v.shape)
        if not np.array_equal(got, v.astype(got.dtype)):
            bad = np.argwhere(got != v.astype(got.dtype))
            raise AssertionError("%s: state field %s differs at %s: got %r want %r" %
                                 (where, k, bad[:4].tolist(), got[tuple(bad[0])], v[tuple(bad[0])]))


@pytest.mark.parametrize("name", gu.TRAJECTORY_FIXTURES + gu.HER_FIXTURES)
def test_oracle_replays_reference_bit_exact(oracle, name):
    fx = gu.load(name)
    N = int(fx["meta_N"])
    cfg = oracle.default_config(1, N, rng_mode=oracle.RNG_MT19937, precision=oracle.POS_EXACT, auto_reset=0,
                                **gu.config_overrides(fx))
    env = oracle.OracleBatch(cfg, oracle.MATH_LIBM)
    if name in gu.HER_FIXTURES:
        env.reset()      # SingleAircraftDiscrete9HEREnv.__init__ calls self.reset() (Discrete9HEREnv.py:32)
    obs0 = env.reset()
    assert np.array_equal(obs0[0], fx["obs0"]), "reset obs differs"
    if "st0_int_x" in fx:
        _assert_state(env.get_state(), gu.prefixed_state(fx, "st0_"), "after reset")
    inj = gu.prefixed_state(fx, "inj_")
    if inj:
        env.set_state(inj)
    actions = fx["actions"]
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    st_steps = {int(t): i for i, t in enumerate(fx["st_steps"])}
    for t in range(len(actions)):
        obs, rew, done, info = env.step(actions[t:t + 1])
        where = "%s step %d" % (name, t)
        assert np.array_equal(obs[0], fx["obs"][t]), where + ": obs"
        assert rew[0] == fx["reward"][t], where + ": reward %r vs %r" % (rew[0], fx["reward"][t])
        assert done[0] == fx["done"][t], where + ": done"
        assert info[0] == fx["info"][t], where + ": info"
        st = env.get_state()
        assert st["no_conflict"][0] == fx["no_conflict"][t], where + ": no_conflict"
        if "nearest" in fx:      # Discrete3HEREnv.step's 4th return value
            nd, nf = env.nearest()
            assert nd[0] == fx["nearest"][t] and nf[0] == fx["nearest_f64"][t], where + ": nearest %r vs %r" % (nd[0], fx["nearest"][t])
        if done[0]:
            assert t in reset_steps
            ro = env.reset()
            assert np.array_equal(ro[0], fx["reset_obs"][reset_steps[t]]), where + ": reset obs"
            st = env.get_state()
        assert np.array_equal(st["int_conflict"][0], fx["conflict"][t]), where + ": conflict flags"
        assert np.array_equal(st["int_promoted"][0], fx["promoted"][t]), where + ": promoted flags"
        if t in st_steps:
            _assert_state(st, gu.prefixed_state(fx, "st_", st_steps[t]), where)
    env.close()


@pytest.mark.parametrize("name", ["g2_goalseek_n8_seed3", "g9_randintruder_n8_seed13"])
def test_oracle_auto_reset_matches_vecenv_order(oracle, name):
    """auto_reset=1: the obs of a done env is its reset obs (dummy_vec_env.py:52-55)."""
    fx = gu.load(name)
    N = int(fx["meta_N"])
    cfg = oracle.default_config(1, N, rng_mode=oracle.RNG_MT19937, precision=oracle.POS_EXACT, auto_reset=1,
                                **gu.config_overrides(fx))
    env = oracle.OracleBatch(cfg, oracle.MATH_LIBM)
    env.reset()
    reset_steps = {int(t): i for i, t in enumerate(fx["reset_steps"])}
    assert reset_steps, "fixture has no terminal step"
    for t in range(len(fx["actions"])):
        obs, rew, done, info = env.step(fx["actions"][t:t + 1])
        want = fx["reset_obs"][reset_steps[t]] if t in reset_steps else fx["obs"][t]
        assert np.array_equal(obs[0], want), "step %d" % t
        assert rew[0] == fx["reward"][t] and done[0] == fx["done"][t] and info[0] == fx["info"][t]


def test_oracle_shared_stream_matches_g10(oracle):
    """B env objects sharing the global np.random, stepped in DummyVecEnv order."""
    fx = gu.load("g10_shared_stream_b4_n8_seed5")
    B, N = int(fx["meta_B"]), int(fx["meta_N"])
    cfg = oracle.default_config(B, N, rng_mode=oracle.RNG_MT19937, precision=oracle.POS_EXACT, auto_reset=1,
                                **gu.config_overrides(fx))
    env = oracle.OracleBatch(cfg, oracle.MATH_LIBM, shared_stream=True)
    obs0 = env.reset()
    assert np.array_equal(obs0, fx["obs0"])
    for t in range(fx["actions"].shape[0]):
        obs, rew, done, info = env.step(fx["actions"][t])
        assert np.array_equal(obs, fx["obs_after"][t]), "step %d" % t
        assert np.array_equal(rew, fx["reward"][t])
        assert np.array_equal(done, fx["done"][t])
        assert np.array_equal(info, fx["info"][t])


@pytest.mark.parametrize("sparse", [False, True])
def test_oracle_her_relabel_matches_reference(oracle, sparse):
    """compute_reward / compute_input_reward of both HER envs (g15), bit-exact float64."""
    fx = gu.load("g15_her_relabel")
    cfg = oracle.default_config(1, int(fx["meta_N"]), obs_mode=oracle.OBS_KNN5, knn=int(fx["meta_knn"]), sparse_reward=int(sparse))
    X = fx["inputs"].astype(np.float64)
    tag = "sparse" if sparse else "dense"
    got = oracle.her_input_reward(cfg, X)
    assert np.array_equal(got, fx["input_reward9_" + tag]) and np.array_equal(got, fx["input_reward3_" + tag])
    got = oracle.her_goal_reward(cfg, X[:, :2], X[:, -2:])
    assert np.array_equal(got, fx["goal_reward9_" + tag]) and np.array_equal(got, fx["goal_reward3_" + tag])
    kinds = set(np.round(fx["input_reward9_sparse"], 6).tolist())
    assert kinds == {-1.0, -0.5, -0.001, 1.0}, kinds     # NMAC, conflict, step, goal rows all present


def test_numpy_legacy_rng_known_answers(oracle):
    """G7: MT19937 seeding, rk_double, polar rk_gauss (with cache) and get_state()."""
    fx = gu.load("g7_rng_kat")
    for s in fx["meta_seeds"]:
        s = int(s)
        assert np.array_equal(oracle.np_random_sample(s, 64), fx["random_sample_%d" % s])
        assert np.array_equal(oracle.np_normal(s, 64, 0.0, 0.5), fx["normal_%d" % s])
        key, pos = oracle.np_state(s, 700)
        assert np.array_equal(key, fx["state_key_%d" % s]) and pos == int(fx["state_pos_%d" % s])


def test_fixtures_cover_the_edge_cases():
    """The committed fixtures must actually exercise what they claim."""
    g2 = gu.load("g2_goalseek_n32_seed1")
    counts = np.bincount(g2["info"], minlength=4)
    assert counts[1] > 0 and counts[2] > 0 and counts[3] > 0, counts
    assert g2["promoted"].any(), "no float64-promoted intruder in G2 (G6 coverage)"
    g3 = gu.load("g3_nmac_early_exit")
    assert g3["info"][0] == 3 and g3["done"][0] == 1
    # intruders after the NMAC index kept their injected positions (terminal obs, normalised by 800)
    N = int(g3["meta_N"])
    for i in (5, 6):
        assert g3["obs"][0][4 * i] == np.float32(g3["inj_int_x"][i]) / np.float32(800)
    assert g3["no_conflict"][0] == 2  # intruder 1 (conflict) and intruder 3 (NMAC) counted, 5 not
    g4 = gu.load("g4_respawn_in_conflict")
    assert g4["info"][0] == 1 and g4["no_conflict"][0] == 1 and g4["conflict"][0][2] == 0
    g5 = gu.load("g5_conflict_masks_goal")
    assert g5["info"][0] == 1 and g5["done"][0] == 0 and 2 in g5["info"]
    g80 = gu.load("g2_goalseek_n80_seed4")
    assert int(g80["meta_N"]) == 80 and N == 8
