#!/usr/bin/env python3
"""Development tool: randomized MCTS configurations, HIP search / move vs the CPU oracle (Philox, portable math)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import uamenv
from oracle import oracle

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    model = int(rs.randint(2))
    w = 4 if model else 6
    N = int(rs.choice([1, 2, 3, 8, 20, 63, 64, 65, 100, 129, 200]))
    sims = int(rs.choice([1, 5, 30, 100, 126, 127, 200, 254]))
    depth = int(rs.randint(1, 6))
    frames = int(rs.choice([1, 3, 10, 16, 33, 63]))
    over = dict(simulations=sims, search_depth=depth, simulate_frame=frames, model=model,
                position_sigma=float(rs.choice([0.0, 0.3])), speed_sigma=float(rs.choice([0.0, 0.07])),
                heading_change_prob=float(rs.choice([0.0, 0.1, 0.5, 1.0])))
    B = int(rs.randint(1, 6))
    roots = np.zeros((B, w * N + 8))
    for b in range(B):
        hd = rs.uniform(0, 2 * np.pi, N); sp = rs.uniform(5 / 3, 8 / 3, N)
        cols = [rs.uniform(0, 800, N), rs.uniform(0, 800, N), sp * np.cos(hd), sp * np.sin(hd)] + ([sp, hd] if w == 6 else [])
        roots[b, :w * N] = np.stack(cols, axis=1).ravel()
        oh = rs.uniform(0, 2 * np.pi)
        roots[b, w * N:] = [rs.uniform(50, 750), rs.uniform(50, 750), 2 * np.cos(oh), 2 * np.sin(oh), 2.0, oh, rs.uniform(0, 800), rs.uniform(0, 800)]
    planner = uamenv.BatchedMCTS(N, seed=int(rs.randint(1 << 30)), **over)
    ocfg = oracle.UamMctsConfig()
    for name, _ in oracle.UamMctsConfig._fields_:
        setattr(ocfg, name, getattr(planner.cfg, name))
    rid = int(rs.randint(1 << 20))
    acts, cq, cn = planner.best_action(roots, root_id0=rid, want_children=True)
    ok = True
    for k in range(B):
        o = oracle.mcts_best_action(ocfg, roots[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=rid + k)
        ok &= np.array_equal(cq[k].cpu().numpy(), o["child_q"]) and np.array_equal(cn[k].cpu().numpy(), o["child_n"]) \
            and np.array_equal(acts[k].cpu().numpy(), o["best_action"])
    a = rs.randint(0, 3, size=(B, 2)).astype(np.int32)
    nxt, fl, rw = planner.move(roots, a, root_id0=rid, sim=3, move=1)
    for k in range(B):
        on, of, orw, _ = oracle.mcts_move(ocfg, roots[k], a[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE,
                                         root_id=rid + k, sim=3, move=1)
        ok &= np.array_equal(nxt[k].cpu().numpy(), on) and np.array_equal(fl[k].cpu().numpy(), of) and rw[k].item() == orw
    if not ok:
        bad += 1
        print("FAIL", N, B, over)
    n += 1
print("fuzz mcts: %d configurations, %d failures, %.0f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
