#!/usr/bin/env python3
"""Development tool: latency of the B=1 gym drop-in classes (NumPy in/out, MT19937 exact mode)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uamenv
for N in (4, 32, 80):
    env = uamenv.SingleAircraftEnv(num_intruders=N)
    env.seed(0)
    env.reset()
    for t in range(50):
        env.step(t % 9)
    T = 1000
    t0 = time.perf_counter()
    for t in range(T):
        ob, r, d, info = env.step(t % 9)
        if d:
            env.reset()
    dt = (time.perf_counter() - t0) / T
    print(json.dumps({"class": "SingleAircraftEnv", "N": N, "us_per_step": round(dt * 1e6, 1), "steps_per_s": round(1 / dt)}))
    env.close()
