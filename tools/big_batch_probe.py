#!/usr/bin/env python3
"""Development probe: step time at a large batch (default 1 M envs x 32: 1.5 GB per step, not cache-resident), best and
median of several 48-step regions.  usage: big_batch_probe.py [B] ; combine with UAMENV_BLOCK=128|256|512"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uamenv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = uamenv.BatchedSingleAircraftEnv(B, 32, seed=0)
env.reset()
acts = [env.policy_actions("random", seed=i + 1).clone() for i in range(8)]
for i in range(24):
    env.step(acts[i % 8])
torch.cuda.synchronize()
ts = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(48):
        env.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 48 * 1e3)
ts.sort()
nbytes = B * (40 * 32 + 150)
print("B=%d block=%s: best %.1f us (%.2f TB/s = %.3f of 8), median %.1f us (%.3f)" %
      (B, os.environ.get("UAMENV_BLOCK", "256"), ts[0], nbytes / ts[0] / 1e6, nbytes / ts[0] / 1e6 / 8, ts[len(ts) // 2], nbytes / ts[len(ts) // 2] / 1e6 / 8))
