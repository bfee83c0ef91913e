#!/usr/bin/env python3
"""Development tool: randomized uam_step_many (single-launch loop or queued launches) vs K separate steps, plus a
get_state/set_state round trip into a fresh handle in the middle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uamenv
import parity_util as pu

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    variant = rs.choice(["discrete9", "continuous", "mcts", "randintruder", "random_start", "her9", "her3"])
    rng = rs.choice(["philox", "philox", "philox", "mt19937"])
    prec = rs.choice(["f32", "f32", "exact"])
    N = int(rs.choice([1, 2, 4, 5, 8, 9, 16, 31, 32, 33, 64, 100, 256]))
    if variant in ("her9", "her3") and N < 6:
        continue
    B = int(rs.choice([1, 3, 32, 33, 257, 1000, 5000, 30000]))
    K = int(rs.choice([1, 2, 5, 16, 17, 40]))
    kw = dict(variant=variant, rng=rng, precision=prec, seed=int(rs.randint(1 << 30)), monitor=bool(rs.randint(2)))
    a = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
    b = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
    a.reset(); b.reset()
    ok = True
    for rep in range(2):
        tape = torch.stack([a.policy_actions("goal" if rs.randint(2) else "random", seed=int(rs.randint(1 << 20))) for _ in range(K)])
        obs_m, rew_m, done_m, info_m = a.step_many(tape)
        for k in range(K):
            o, r, d, i = b.step(tape[k])
            ok &= torch.equal(rew_m[k], r) and torch.equal(done_m[k], d.view(torch.uint8)) and torch.equal(info_m[k], i.codes)
        ok &= torch.equal(obs_m, o)
        if rep == 0:      # checkpoint a, restore into a fresh handle, continue with that one
            st = a.get_state()
            c = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
            c.reset()
            c.set_state(st)
            a.close()
            a = c
    try:
        pu.compare_states(a.get_state(), b.get_state(), "end", hold_sh=variant == "randintruder")
    except AssertionError as ex:
        ok = False
        print(ex)
    if not ok:
        bad += 1
        print("FAIL B=%d N=%d K=%d %s" % (B, N, K, kw))
    a.close(); b.close()
    n += 1
print("fuzz step_many: %d configurations, %d failures, %.0f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
