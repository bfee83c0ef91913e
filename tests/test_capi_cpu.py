"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/uamenv.h declares, and its no-GPU behaviour is a loud failure, not a fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import uamenv
    uamenv.build()
    return uamenv._capi


def declared_functions():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("uamenv.h", "uamenv_bench.h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uam_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(capi):
    L = C.CDLL(capi.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 18, names
    for n in names:
        assert hasattr(L, n), "libuamenv_hip.so does not export %s" % n
    assert sorted(s[0] for s in capi.SYMBOLS) == names, "python binding and header disagree"


def test_config_struct_matches_header_and_reference_constants(capi):
    cfg = capi.default_config(3, 7)
    assert cfg.struct_size == C.sizeof(capi.UamConfig)
    # Simulators/config.py:16-31 values
    assert cfg.minimum_separation == 18.5 and cfg.nmac_dist == 5.0 and cfg.initial_min_dist == 100.0
    assert cfg.goal_radius == 20.0 and cfg.min_speed == 50 / 30 and cfg.max_speed == 80 / 30
    assert cfg.d_heading == np.radians(5) and cfg.heading_sigma == np.radians(4)
    assert cfg.nmac_penalty == -1.0 and cfg.conflict_penalty == -0.5 and cfg.goal_reward == 1.0
    assert capi.obs_dim(cfg) == 4 * 7 + 8
    cfg.obs_mode = capi.OBS_RAW6
    assert capi.obs_dim(cfg) == 6 * 7 + 8


def test_oracle_and_product_config_layouts_agree(capi, oracle):
    """The oracle restates uam_config_default on its own; both must agree byte for byte."""
    a = capi.default_config(5, 9)
    b = oracle.default_config(5, 9)
    assert C.sizeof(a) == C.sizeof(b)
    assert bytes(a) == bytes(b)


def test_no_gpu_is_a_loud_failure(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import uamenv
    cfg = capi.default_config(4, 8)
    h = C.c_void_p()
    rc = capi.lib().uam_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -19 and b"no CPU backend" in capi.lib().uam_last_error()
    with pytest.raises(RuntimeError):
        uamenv.BatchedSingleAircraftEnv(4, 8)
    with pytest.raises(RuntimeError):
        uamenv.SingleAircraftEnv(4)


def test_bad_config_is_rejected_before_touching_a_device(capi):
    h = C.c_void_p()
    for field, val in [("num_envs", 0), ("num_intruders", -1), ("num_intruders", 257), ("obs_mode", 9),
                       ("rng_mode", 5), ("struct_size", 8)]:
        cfg = capi.default_config(4, 8)
        setattr(cfg, field, val)
        rc = capi.lib().uam_create(C.byref(cfg), 0, C.byref(h))
        assert rc == -22, (field, rc)
        assert capi.lib().uam_last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import, include or load it."""
    pkg = os.path.join(ROOT, "gym-guidance-collision-avoidance-single_amd")
    pat = re.compile(r"(^|\n)\s*(from\s+oracle|import\s+oracle|#\s*include\s*[<\"][^\n]*oracle)|liboracle|oracle/_build")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), "%s references the oracle" % os.path.join(dirpath, f)


def test_variant_configs_follow_the_reference_classes(capi):
    """Host logic (no GPU): every variant name maps to the obs/action/dynamics combination of its reference class."""
    import uamenv
    want = {  # variant: (obs_mode, action_mode, dynamics, random_own_start, obs_dim at N=8)
        "discrete9": (0, 0, 0, 0, 40), "continuous": (0, 2, 0, 0, 40), "continuous2": (0, 4, 0, 0, 40), "mcts": (1, 1, 0, 0, 40),
        "randintruder": (2, 1, 1, 0, 56), "random_start": (0, 0, 0, 1, 40), "her9": (3, 0, 0, 1, 28), "her3": (3, 3, 0, 1, 28),
    }
    for variant, (om, am, dyn, ros, D) in want.items():
        cfg = uamenv.make_config(16, 8, variant=variant)
        assert (cfg.obs_mode, cfg.action_mode, cfg.dynamics, cfg.random_own_start) == (om, am, dyn, ros), variant
        assert capi.obs_dim(cfg) == D, variant
    her3 = uamenv.make_config(4, 8, variant="her3")
    assert her3.shaped_reward == 1 and her3.goal_margin == 100.0 and her3.conflict_coeff == 0.00025 and her3.knn == 4
    assert uamenv.make_config(4, 8, variant="her9").shaped_reward == 0

    class Sparse(uamenv.Config):
        sparse_reward = True
    assert uamenv.make_config(4, 8, variant="her9", config=Sparse).wall_terminates == 1      # Discrete9HEREnv.py:217-221
    assert uamenv.make_config(4, 8, variant="discrete9", config=Sparse).wall_terminates == 0
    # the packaged copy's constants (envs/config.py + the literals of its _terminal_reward)
    pk = uamenv.make_config(4, 8, variant="continuous2", config=uamenv.PackagedConfig)
    assert (pk.nmac_penalty, pk.conflict_penalty, pk.wall_penalty, pk.goal_reward, pk.wall_terminates) == (-5, -1, -100, 1, 1)
    assert pk.heading_sigma == np.radians(2) and pk.action_mode == 4
    with pytest.raises(AttributeError):
        uamenv.make_config(4, 8, no_such_field=1)


def test_host_infos_sequences(capi):
    """LazyInfos / HostInfos rebuild the reference's info objects from code arrays (host logic, no GPU)."""
    import uamenv
    codes = np.array([0, 1, 2, 3, 4], dtype=np.uint8)
    done = np.array([False, False, True, True, True])
    h = uamenv.HostInfos(codes, done, np.array([0.0, 0.0, 1.5, -1.0, -0.5]), np.array([0, 0, 120, 7, 33], dtype=np.int32), 12.5)
    assert [i["result"] for i in h] == ["", "c", "g", "n", "w"] and len(h) == 5
    assert "episode" not in h[0] and h[2]["episode"] == {"r": 1.5, "l": 120, "t": 12.5}
    assert h.episodes() == [(2, 1.5, 120), (3, -1.0, 7), (4, -0.5, 33)]
    assert uamenv.HostInfos(codes, done, None, None, 0.0).episodes() == []


def test_bad_planner_config_is_rejected_before_touching_a_device(capi):
    """uam_mcts_best_action / uam_mcts_move validate the configuration first: a bad one is UAM_EINVAL (-22) with a
    message on a box without a GPU too.  simulate_frame is bounded by the wave: one lane per frame of a move plus one
    for the state before it."""
    lib = capi.lib()
    for field, value, word in (("simulate_frame", 64, "simulate_frame"), ("simulate_frame", 0, "simulate_frame"),
                               ("simulations", 255, "simulations"), ("search_depth", 0, "search_depth"),
                               ("num_intruders", 257, "num_intruders"), ("model", 7, "model")):
        cfg = capi.UamMctsConfig()
        assert lib.uam_mcts_config_default(C.byref(cfg), 8) == 0
        setattr(cfg, field, value)
        assert lib.uam_mcts_workspace_bytes(C.byref(cfg), 4) == -1
        assert word in lib.uam_last_error().decode(), (field, lib.uam_last_error())
        rc = lib.uam_mcts_best_action(C.byref(cfg), 0, None, 4, 0, None, None, None, None, None)
        assert rc == -22 and word in lib.uam_last_error().decode()
    cfg = capi.UamMctsConfig()
    assert lib.uam_mcts_config_default(C.byref(cfg), 8) == 0 and cfg.simulate_frame == 10 and cfg.simulations == 100
    assert lib.uam_mcts_workspace_bytes(C.byref(cfg), 4) == 4 * 101 * (5 * 64 + 8) * 8
