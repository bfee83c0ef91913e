#!/usr/bin/env python3
"""Development tool: randomized configurations, HIP env vs CPU oracle in lockstep (tests/parity_util.py)."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import uamenv
from oracle import oracle
import parity_util as pu

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    variant = rs.choice(["discrete9", "continuous", "continuous2", "mcts", "randintruder", "random_start", "her9", "her3"])
    rng = os.environ.get("FUZZ_RNG") or rs.choice(["philox", "philox", "mt19937", "mt19937", "mt19937_shared"])
    prec = rs.choice(["f32", "exact"])
    N = int(rs.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 64, 65, 100, 128, 200, 256]))
    B = int(rs.choice([1, 2, 3, 7, 31, 32, 33, 100, 257, 1000]))
    over = {}
    if variant in ("her9", "her3"):
        if N < 2:
            continue
        over["knn"] = int(rs.randint(1, min(N, 9)))
    class Cfg(uamenv.Config):
        sparse_reward = bool(rs.randint(2))
    T = int(rs.randint(20, 120)) if rng == "mt19937" else int(rs.randint(50, 300))
    kw = dict(variant=variant, rng=rng, precision=prec, config=Cfg, auto_reset=bool(rs.randint(4) > 0), **over)
    desc = "B=%d N=%d T=%d %s sparse=%s" % (B, N, T, {k: v for k, v in kw.items() if k != "config"}, Cfg.sparse_reward)
    try:
        pu.run_lockstep(uamenv, oracle, B, N, T, policy=rs.choice(["random", "goal", "mixed"]), seed=int(rs.randint(1 << 30)),
                        check_state_every=7, reward_tol=1e-6 if variant == "her3" else 0.0, **kw)
    except Exception:
        bad += 1
        print("FAIL", desc); traceback.print_exc(limit=2)
    n += 1
print("fuzz: %d configurations, %d failures, %.0f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
