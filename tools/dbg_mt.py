import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import uamenv
from oracle import oracle
import parity_util as pu
cases = [("discrete9", 256, 80, "exact"), ("discrete9", 256, 80, "f32"), ("her3", 256, 80, "f32"), ("discrete9", 64, 256, "exact"), ("randintruder", 64, 64, "exact"), ("randintruder", 32, 120, "exact")]
for variant, B, N, prec in cases:
    try:
        out = pu.run_lockstep(uamenv, oracle, B, N, 120, policy="goal", rng="mt19937", precision=prec, variant=variant, check_state_every=4,
                              reward_tol=1e-6 if variant == "her3" else 0.0)
        print(os.environ.get("UAMENV_MT_WIN"), variant, B, N, prec, "ok", out["spawns"], out["counts"].tolist())
    except AssertionError as e:
        print(os.environ.get("UAMENV_MT_WIN"), variant, B, N, prec, "FAIL", str(e)[:120])
