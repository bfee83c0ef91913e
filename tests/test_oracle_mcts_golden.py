"""Pin the oracle's MCTS forward model / UCT search against the reference's own outputs
(tests/golden/g11_*, g12_*: nodes_single_randintru.py move(), search_single.py best_action(), recorded
after np.random.seed(s)).  Bit-exact float64 states, flags, rewards, child statistics and even the
position of the MT19937 stream afterwards."""
import numpy as np
import pytest

import golden_util as gu

MOVE_FIXTURES = ["g11_mcts_move_n8", "g11_mcts_move_n20", "g11_mcts_move_n8_flags", "g11_mcts_move_n8_noisy",
                 # the 4-wide model of Algorithms/MCTS/nodes_single.py (Agent.py's planner)
                 "g16_mcts4_move_n8", "g16_mcts4_move_n20", "g16_mcts4_move_n8_flags", "g16_mcts4_move_n8_noisy",
                 # every sigma zero: the cases the GPU is compared with DIRECTLY (tests/test_gpu_mcts.py)
                 "g18_mcts4_move_zero_n8", "g18_mcts4_move_zero_n20", "g18_mcts4_move_zero_n8_flags"]
SEARCH_FIXTURES = ["g12_mcts_search_n8", "g12_mcts_search_n20", "g12_mcts_search_n8_flags",
                   "g17_mcts4_search_n8", "g17_mcts4_search_n20"]


def _model(fx):
    return 1 if int(fx.get("meta_width", 6)) == 4 else 0


def _cfg(oracle, fx, name):
    over = {}
    if name.endswith("noisy"):
        over = dict(position_sigma=10 / 30, speed_sigma=2 / 30)
    if "_zero_" in name:
        over = dict(position_sigma=0.0, speed_sigma=0.0, heading_sigma=0.0)
    return oracle.mcts_config(int(fx["meta_N"]), model=_model(fx), **over)


@pytest.mark.parametrize("name", MOVE_FIXTURES)
def test_move_matches_reference_bit_exact(oracle, name):
    fx = gu.load(name)
    cfg = _cfg(oracle, fx, name)
    for k in range(len(fx["seeds"])):
        nxt, flags, rew, pos = oracle.mcts_move(cfg, fx["states"][k], fx["actions"][k], np_seed=int(fx["seeds"][k]))
        assert np.array_equal(flags, fx["flags"][k]), (name, k, flags, fx["flags"][k])
        assert np.array_equal(nxt, fx["next_states"][k]), (name, k, np.abs(nxt - fx["next_states"][k]).max())
        assert rew == fx["rewards"][k], (name, k)
        assert pos == int(fx["rng_pos"][k]), (name, k, "stream position")
    if name.endswith("flags"):
        assert fx["flags"].sum(0).min() > 0, "fixture must exercise hit_wall, conflict and reach_goal"
    if name == "g16_mcts4_move_n8_flags":
        # upstream quirk: the last intruder is never moved ((len - 9) // 4 = N - 1)
        N = int(fx["meta_N"])
        assert np.array_equal(fx["next_states"][:, 4 * (N - 1):4 * N], fx["states"][:, 4 * (N - 1):4 * N])


@pytest.mark.parametrize("name", SEARCH_FIXTURES)
def test_best_action_matches_reference_bit_exact(oracle, name):
    fx = gu.load(name)
    cfg = oracle.mcts_config(int(fx["meta_N"]), simulations=int(fx["meta_sims"]), search_depth=int(fx["meta_depth"]),
                             model=_model(fx))
    for k in range(len(fx["seeds"])):
        out = oracle.mcts_best_action(cfg, fx["states"][k], np_seed=int(fx["seeds"][k]))
        assert np.array_equal(out["best_action"], fx["best_action"][k]), (name, k)
        assert np.array_equal(out["child_action"], fx["child_action"][k]), (name, k)
        assert np.array_equal(out["child_n"], fx["child_n"][k]), (name, k)
        assert np.array_equal(out["child_q"], fx["child_q"][k]), (name, k, out["child_q"] - fx["child_q"][k])
        assert out["root_n"] == fx["root_n"][k] and out["n_nodes"] == int(fx["n_nodes"][k])
        assert out["rng_pos"] == int(fx["rng_pos"][k])


def test_philox_search_is_deterministic_and_seed_sensitive(oracle):
    fx = gu.load("g12_mcts_search_n8")
    cfg = oracle.mcts_config(8, seed=11)
    a = oracle.mcts_best_action(cfg, fx["states"][0], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=3)
    b = oracle.mcts_best_action(cfg, fx["states"][0], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=3)
    c = oracle.mcts_best_action(cfg, fx["states"][0], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=4)
    assert np.array_equal(a["child_q"], b["child_q"]) and not np.array_equal(a["child_q"], c["child_q"])
    assert a["n_nodes"] == 101 and a["root_n"] == 100.0
