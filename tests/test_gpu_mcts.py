"""Batched MCTS on the GPU (through the C-ABI) against the CPU oracle on the same inputs: the oracle's
MT19937/libm mode is pinned bit-exactly to the reference (tests/test_oracle_mcts_golden.py); here its
Philox/portable-math mode must agree with the HIP kernels to the bit — next states, flags, rewards, the
chosen action and the root children's (q, n)."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uamenv():
    import torch
    assert torch.cuda.is_available()
    import uamenv as m
    m._capi.lib()
    return m


def _ocfg(oracle, cfg):
    o = oracle.UamMctsConfig()
    for name, _ in oracle.UamMctsConfig._fields_:
        setattr(o, name, getattr(cfg, name))
    return o


def _roots(name, extra_noise=0):
    fx = gu.load(name)
    roots = np.unique(fx["states"], axis=0)
    return np.ascontiguousarray(roots)


@pytest.mark.parametrize("name,over", [("g11_mcts_move_n8", {}), ("g11_mcts_move_n20", {}), ("g11_mcts_move_n8_flags", {}),
                                       ("g11_mcts_move_n8", dict(position_sigma=10 / 30, speed_sigma=2 / 30))])
def test_move_matches_oracle_bit_exact(uamenv, oracle, name, over):
    roots = _roots(name)
    N = (roots.shape[1] - 8) // 6
    planner = uamenv.BatchedMCTS(N, seed=99, **over)
    ocfg = _ocfg(oracle, planner.cfg)
    states = np.repeat(roots, 9, axis=0)
    actions = np.tile(np.array([(a, b) for a in range(3) for b in range(3)], dtype=np.int32), (len(roots), 1))
    nxt, flags, rew = planner.move(states, actions, root_id0=40, sim=7, move=2)
    nxt, flags, rew = nxt.cpu().numpy(), flags.cpu().numpy(), rew.cpu().numpy()
    seen = np.zeros(3, int)
    for k in range(len(states)):
        on, of, orw, _ = oracle.mcts_move(ocfg, states[k], actions[k], rng_mode=oracle.RNG_PHILOX,
                                         math_impl=oracle.MATH_PORTABLE, root_id=40 + k, sim=7, move=2)
        assert np.array_equal(flags[k], of), (k, flags[k], of)
        assert np.array_equal(nxt[k], on), (k, np.abs(nxt[k] - on).max())
        assert rew[k] == orw
        seen += of
    if name.endswith("flags"):
        assert seen.min() > 0


@pytest.mark.parametrize("name,sims,depth", [("g12_mcts_search_n8", 100, 3), ("g12_mcts_search_n20", 100, 3),
                                             ("g12_mcts_search_n8_flags", 100, 3), ("g12_mcts_search_n8", 40, 2),
                                             ("g12_mcts_search_n8", 254, 4)])
def test_best_action_matches_oracle_bit_exact(uamenv, oracle, name, sims, depth):
    roots = gu.load(name)["states"]
    N = (roots.shape[1] - 8) // 6
    planner = uamenv.BatchedMCTS(N, seed=5, simulations=sims, search_depth=depth)
    ocfg = _ocfg(oracle, planner.cfg)
    acts, cq, cn = planner.best_action(roots, root_id0=1000, want_children=True)
    acts, cq, cn = acts.cpu().numpy(), cq.cpu().numpy(), cn.cpu().numpy()
    for k in range(len(roots)):
        o = oracle.mcts_best_action(ocfg, roots[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE,
                                    root_id=1000 + k)
        assert np.array_equal(cn[k], o["child_n"]), (k, cn[k], o["child_n"])
        assert np.array_equal(cq[k], o["child_q"]), (k, cq[k] - o["child_q"])
        assert np.array_equal(acts[k], o["best_action"]), (k, acts[k], o["best_action"])


def test_bad_mcts_config_is_rejected(uamenv):
    with pytest.raises(uamenv.UamError):
        uamenv.BatchedMCTS(8, simulations=255).best_action(np.zeros((1, 56)))
    with pytest.raises(uamenv.UamError):
        uamenv.BatchedMCTS(300).best_action(np.zeros((1, 6 * 300 + 8)))
    with pytest.raises(uamenv.UamError):      # the reference raises too: np.argmax([]) on a terminal root
        uamenv.BatchedMCTS(8, search_depth=0).best_action(np.zeros((1, 56)))
    with pytest.raises(uamenv.UamError):      # one lane per frame of a move plus one for the state before it
        uamenv.BatchedMCTS(8, simulate_frame=64).best_action(np.zeros((1, 56)))
    with pytest.raises(ValueError):
        uamenv.BatchedMCTS(8).best_action(np.zeros((1, 40)))


def test_long_moves_and_overflowing_change_queue(uamenv, oracle):
    """63 frames per move (the lane-walked recurrences at their limit) and a heading-change rate that overflows the
    change queue (more than 128 changes in a move: the slow frame loop), both against the oracle."""
    rs = np.random.RandomState(21)
    for N, over in ((20, dict(simulate_frame=63, simulations=20, search_depth=2)),
                    (64, dict(heading_change_prob=0.5, simulations=30, search_depth=2)),
                    (40, dict(heading_change_prob=1.0, simulate_frame=16, simulations=12, search_depth=3, speed_sigma=0.07))):
        B = 3
        roots = np.zeros((B, 6 * N + 8))
        for b in range(B):
            hd = rs.uniform(0, 2 * np.pi, N)
            sp = rs.uniform(5 / 3, 8 / 3, N)
            it = np.stack([rs.uniform(0, 800, N), rs.uniform(0, 800, N), sp * np.cos(hd), sp * np.sin(hd), sp, hd], axis=1)
            roots[b, :6 * N] = it.ravel()
            roots[b, 6 * N:] = [400, 400, 1.2, 1.2, 1.7, np.pi / 4, rs.uniform(0, 800), rs.uniform(0, 800)]
        planner = uamenv.BatchedMCTS(N, seed=17, **over)
        ocfg = _ocfg(oracle, planner.cfg)
        acts, cq, cn = planner.best_action(roots, root_id0=5, want_children=True)
        for k in range(B):
            o = oracle.mcts_best_action(ocfg, roots[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=5 + k)
            assert np.array_equal(cq[k].cpu().numpy(), o["child_q"]) and np.array_equal(cn[k].cpu().numpy(), o["child_n"])
            assert np.array_equal(acts[k].cpu().numpy(), o["best_action"])
        a = rs.randint(0, 3, size=(B, 2)).astype(np.int32)
        nxt, fl, rw = planner.move(roots, a, root_id0=5, sim=2, move=1)
        for k in range(B):
            on, of, orw, _ = oracle.mcts_move(ocfg, roots[k], a[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE,
                                             root_id=5 + k, sim=2, move=1)
            assert np.array_equal(nxt[k].cpu().numpy(), on) and np.array_equal(fl[k].cpu().numpy(), of) and rw[k].item() == orw


def test_many_intruders_and_batch_invariance(uamenv, oracle):
    """N=150 (3 intruders per lane) against the oracle; and a root's result must not depend on its batch."""
    rs = np.random.RandomState(3)
    N, B = 150, 6
    roots = np.zeros((B, 6 * N + 8))
    for b in range(B):
        hd = rs.uniform(0, 2 * np.pi, N)
        sp = rs.uniform(5 / 3, 8 / 3, N)
        it = np.stack([rs.uniform(0, 800, N), rs.uniform(0, 800, N), sp * np.cos(hd), sp * np.sin(hd), sp, hd], axis=1)
        roots[b, :6 * N] = it.ravel()
        roots[b, 6 * N:] = [400, 400, 1.2, 1.2, 1.7, np.pi / 4, rs.uniform(0, 800), rs.uniform(0, 800)]
    planner = uamenv.BatchedMCTS(N, seed=8, simulations=30, search_depth=2)
    ocfg = _ocfg(oracle, planner.cfg)
    acts, cq, cn = planner.best_action(roots, root_id0=0, want_children=True)
    for k in range(B):
        o = oracle.mcts_best_action(ocfg, roots[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=k)
        assert np.array_equal(cq[k].cpu().numpy(), o["child_q"]) and np.array_equal(acts[k].cpu().numpy(), o["best_action"])
    a2, q2, n2 = planner.best_action(roots[3:5], root_id0=3, want_children=True)
    assert np.array_equal(q2.cpu().numpy(), cq[3:5].cpu().numpy())


def test_roots_from_env_feed_the_planner(uamenv):
    """Agent_RandInt.py loop shape: env obs -> root state -> best action -> env.step, all on the device."""
    import torch
    B, N = 64, 8
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant="randintruder", seed=2)
    obs = env.reset()
    planner = uamenv.BatchedMCTS(N, seed=2, simulations=30)
    roots = planner.roots_from_env(env)
    assert roots.shape == (B, 6 * N + 8) and roots.dtype == torch.float64
    assert torch.allclose(roots.float(), obs, atol=1e-4)       # same numbers as the raw observation
    for t in range(3):
        acts = planner.best_action(planner.roots_from_env(env), root_id0=t * B)
        assert bool(((acts >= 0) & (acts <= 2)).all())
        env.step(acts)
    env.close()


# ---------------------------------------------------------------- the 4-wide model (Algorithms/MCTS/nodes_single.py)
@pytest.mark.parametrize("name,over", [("g16_mcts4_move_n8", {}), ("g16_mcts4_move_n20", {}), ("g16_mcts4_move_n8_flags", {}),
                                       ("g16_mcts4_move_n8", dict(position_sigma=10 / 30, speed_sigma=2 / 30))])
def test_straight4_move_matches_oracle_bit_exact(uamenv, oracle, name, over):
    roots = _roots(name)
    N = (roots.shape[1] - 8) // 4
    planner = uamenv.BatchedMCTS(N, seed=99, model="straight4", **over)
    assert planner.state_dim == 4 * N + 8
    ocfg = _ocfg(oracle, planner.cfg)
    states = np.repeat(roots, 9, axis=0)
    actions = np.tile(np.array([(a, b) for a in range(3) for b in range(3)], dtype=np.int32), (len(roots), 1))
    nxt, flags, rew = planner.move(states, actions, root_id0=40, sim=7, move=2)
    nxt, flags, rew = nxt.cpu().numpy(), flags.cpu().numpy(), rew.cpu().numpy()
    seen = np.zeros(3, int)
    for k in range(len(states)):
        on, of, orw, _ = oracle.mcts_move(ocfg, states[k], actions[k], rng_mode=oracle.RNG_PHILOX,
                                         math_impl=oracle.MATH_PORTABLE, root_id=40 + k, sim=7, move=2)
        assert np.array_equal(flags[k], of), (k, flags[k], of)
        assert np.array_equal(nxt[k], on), (k, np.abs(nxt[k] - on).max())
        assert rew[k] == orw
        seen += of
    assert np.array_equal(nxt[:, 4 * (N - 1):4 * N], states[:, 4 * (N - 1):4 * N]), "the last intruder is never moved (upstream quirk)"
    if name.endswith("flags"):
        assert seen.min() > 0


@pytest.mark.parametrize("name", ["g18_mcts4_move_zero_n8", "g18_mcts4_move_zero_n20", "g18_mcts4_move_zero_n8_flags"])
def test_straight4_zero_noise_move_matches_the_reference_directly(uamenv, name):
    """No oracle in between: with every sigma zero SingleAircraftState.move (nodes_single.py:39-100) is a
    deterministic function of (state, action), so uam_mcts_move — Philox stream, portable math — must reproduce what
    the REFERENCE returned (fixture recorded by make_golden_mcts.py): flags bit-exact, next state and reward to the
    rounding of libm-vs-portable sincos / pow (<= 1e-10 on coordinates up to 800)."""
    fx = gu.load(name)
    N = int(fx["meta_N"])
    planner = uamenv.BatchedMCTS(N, seed=123, model="straight4", position_sigma=0.0, speed_sigma=0.0, heading_sigma=0.0)
    nxt, flags, rew = planner.move(fx["states"], fx["actions"], root_id0=7, sim=3, move=1)
    nxt, flags, rew = nxt.cpu().numpy(), flags.cpu().numpy(), rew.cpu().numpy()
    assert np.array_equal(flags, fx["flags"]), np.argwhere(flags != fx["flags"])[:4]
    d = np.abs(nxt - fx["next_states"])
    assert d.max() <= 1e-10, (d.max(), np.unravel_index(d.argmax(), d.shape))
    assert np.abs(rew - fx["rewards"]).max() <= 1e-12
    # and it does not depend on the random stream at all
    n2, f2, r2 = uamenv.BatchedMCTS(N, seed=9, model="straight4", position_sigma=0.0, speed_sigma=0.0,
                                    heading_sigma=0.0).move(fx["states"], fx["actions"], root_id0=99, sim=0, move=0)
    assert np.array_equal(n2.cpu().numpy(), nxt) and np.array_equal(f2.cpu().numpy(), flags)
    if name.endswith("flags"):
        assert fx["flags"].sum(0).min() > 0


@pytest.mark.parametrize("name,sims,depth", [("g17_mcts4_search_n8", 100, 3), ("g17_mcts4_search_n20", 100, 3),
                                             ("g17_mcts4_search_n8", 200, 4)])
def test_straight4_best_action_matches_oracle_bit_exact(uamenv, oracle, name, sims, depth):
    roots = gu.load(name)["states"]
    N = (roots.shape[1] - 8) // 4
    planner = uamenv.BatchedMCTS(N, seed=5, simulations=sims, search_depth=depth, model="straight4")
    ocfg = _ocfg(oracle, planner.cfg)
    acts, cq, cn = planner.best_action(roots, root_id0=1000, want_children=True)
    acts, cq, cn = acts.cpu().numpy(), cq.cpu().numpy(), cn.cpu().numpy()
    for k in range(len(roots)):
        o = oracle.mcts_best_action(ocfg, roots[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE,
                                    root_id=1000 + k)
        assert np.array_equal(cn[k], o["child_n"]), (k, cn[k], o["child_n"])
        assert np.array_equal(cq[k], o["child_q"]), (k, cq[k] - o["child_q"])
        assert np.array_equal(acts[k], o["best_action"]), (k, acts[k], o["best_action"])


def test_agent_loop_on_mcts_env(uamenv):
    """Agent.py loop shape: SingleAircraftMCTSEnv obs -> 4-wide root state -> best action -> env.step."""
    import torch
    B, N = 64, 8
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant="mcts", seed=2)
    obs = env.reset()
    planner = uamenv.BatchedMCTS(N, seed=2, simulations=30, model="straight4")
    roots = planner.roots_from_env(env)
    assert roots.shape == (B, 4 * N + 8) and roots.dtype == torch.float64
    assert torch.allclose(roots.float(), obs, atol=1e-4)       # same numbers as the raw observation
    with pytest.raises(ValueError):
        uamenv.BatchedMCTS(N).roots_from_env(env)              # 6-wide planner on a 4-wide env
    for t in range(3):
        acts = planner.best_action(planner.roots_from_env(env), root_id0=t * B)
        assert bool(((acts >= 0) & (acts <= 2)).all())
        env.step(acts)
    env.close()


def test_planner_avoids_collisions_end_to_end(uamenv):
    """The Agent_RandInt.py experiment in small: planning with the batched MCTS must cut the NMAC rate of the
    goal-seeking heuristic by a wide margin on the same seeds (functional check of the whole planner stack)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("mcts_agent", os.path.join(os.path.dirname(__file__), "..", "examples", "mcts_agent.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    planned = mod.run("mcts", 512, 20, 600, 100, 3, seed=3)
    greedy = mod.run("goal", 512, 20, 600, 100, 3, seed=3)
    assert planned["episodes"] > 500 and greedy["episodes"] > 500
    assert greedy["nmac_prob"] > 0.05 and planned["nmac_prob"] < 0.25 * greedy["nmac_prob"], (planned, greedy)
    assert planned["goal_prob"] > 0.95
