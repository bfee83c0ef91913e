#!/usr/bin/env python3
"""Development tool: PCIe-inclusive rate of the NumPy-facing VecEnv adapter (never bench.py's `value`)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import uamenv
for B, N, T in ((4096, 8, 400), (65536, 32, 100)):
    venv = uamenv.UamVecEnv(B, N, seed=0, monitor=True)
    venv.reset()
    acts = np.random.RandomState(0).randint(0, 9, size=(8, B)).astype(np.int32)
    for t in range(20):
        venv.step(acts[t % 8])
    t0 = time.perf_counter()
    for t in range(T):
        obs, rew, done, infos = venv.step(acts[t % 8])
    dt = (time.perf_counter() - t0) / T
    per_step_bytes = obs.nbytes + rew.nbytes + done.nbytes + B + B * 12 + acts[0].nbytes
    print(json.dumps({"B": B, "N": N, "us_per_step": round(dt * 1e6, 1), "env_steps_per_s": round(B / dt),
                      "host_bytes_per_step": per_step_bytes, "pcie_GBps": round(per_step_bytes / dt / 1e9, 2)}))
    venv.close()
