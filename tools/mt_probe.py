#!/usr/bin/env python3
"""Development probe: MT19937 / exact step time right after reset() (no env can finish an episode within ~18 steps:
every intruder spawns >= 100 away) against the steady state (a few dozen in-step resets per launch at 65536 envs).
usage: mt_probe.py [B] [N] [policy]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uamenv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pol = sys.argv[3] if len(sys.argv) > 3 else "random"
env = uamenv.BatchedSingleAircraftEnv(B, N, rng=os.environ.get("PROBE_RNG", "mt19937"), precision=os.environ.get("PROBE_PRECISION", "exact"), seed=0,
                                      auto_reset=os.environ.get("PROBE_AUTO_RESET", "1") == "1")
acts = [env.policy_actions(pol, seed=i + 1).clone() for i in range(16)]


def timed(n, label):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nd = 0
    for i in range(n):
        _, _, d, _ = env.step(acts[i % 16])
    e1.record(); torch.cuda.synchronize()
    print("%-28s %.2f us/step   (dones in the last step: %d)" % (label, e0.elapsed_time(e1) / n * 1e3, int(d.sum())))


for rep in range(3):
    env.reset()
    timed(12, "after reset, 12 steps")
for _ in range(400):
    env.step(acts[_ % 16])
for rep in range(3):
    timed(100, "steady state, 100 steps")
env.close()
