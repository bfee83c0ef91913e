"""Helpers shared by the golden-fixture tests (oracle- and GPU-side)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAJECTORY_FIXTURES = [
    "g1_discrete9_n4_seed0", "g2_goalseek_n8_seed3", "g2_goalseek_n32_seed1", "g2_goalseek_n80_seed4",
    "g3_nmac_early_exit", "g4_respawn_in_conflict", "g5_conflict_masks_goal",
    "g8_continuous_n8_seed7", "g8b_packaged_continuous2_n8_seed17", "g8c_packaged_default_n0_seed23", "g9_randintruder_n8_seed13", "g9_randintruder_n64_seed14",
    "g9b_mctsenv_n8_seed11", "g6b_randomenv_n8_seed31",
]

# SingleAircraftDiscrete9HEREnv (k-nearest observation, random ownship start, optional wall termination);
# its __init__ runs one reset() before the recorded one
HER_FIXTURES = ["g13_her9_n8_seed41", "g13_her9_n32_seed42", "g13_her9_sparse_n8_seed43",
                "g14_her3_n8_seed51", "g14_her3_n32_seed52", "g14_her3_sparse_n8_seed53", "g14_her3_straight_n80_seed54"]

STATE_KEYS = ["int_x", "int_y", "int_vx", "int_vy", "int_speed", "int_heading", "int_conflict", "int_promoted",
              "own_x", "own_y", "own_heading", "own_speed", "own_vx", "own_vy", "goal_x", "goal_y", "no_conflict"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def all_fixture_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def config_overrides(fx, **extra):
    """uam_config field overrides encoded in a fixture's meta_* entries."""
    over = dict(obs_mode=int(fx["meta_obs_mode"]), action_mode=int(fx["meta_action_mode"]),
                dynamics=int(fx["meta_dynamics"]), seed=int(fx["meta_seed"]))
    for k in ("heading_sigma", "speed_sigma", "nmac_penalty", "conflict_penalty", "wall_penalty", "goal_reward"):
        if "meta_" + k in fx:
            over[k] = float(fx["meta_" + k])
    if "meta_wall_terminates" in fx:          # the packaged 2Env ends the episode when the ownship leaves the map
        over["wall_terminates"] = int(fx["meta_wall_terminates"])
    if "meta_random_own_start" in fx:
        over["random_own_start"] = int(fx["meta_random_own_start"])
    if int(fx["meta_obs_mode"]) == 3:   # HER family
        sparse = int(fx["meta_sparse_reward"])
        over.update(knn=int(fx["meta_knn"]), random_own_start=1, sparse_reward=sparse, wall_terminates=sparse)
        if "meta_shaped_reward" in fx:   # Discrete3HER
            over.update(shaped_reward=int(fx["meta_shaped_reward"]), goal_margin=float(fx["meta_goal_margin"]),
                        conflict_coeff=float(fx["meta_conflict_coeff"]))
    over.update(extra)
    return over


def prefixed_state(fx, prefix, idx=None):
    """{field: array shaped [1, ...]} from st0_/inj_/st_ entries (idx selects a row of st_*)."""
    st = {}
    for k in STATE_KEYS:
        key = prefix + k
        if key not in fx:
            continue
        v = fx[key] if idx is None else fx[key][idx]
        st[k] = np.ascontiguousarray(np.asarray(v)[None, ...])
    return st
