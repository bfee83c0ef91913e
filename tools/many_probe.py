#!/usr/bin/env python3
"""Development probe: run the looped kernel (uam_rollout, K steps per launch) on one shape a few times; meant to sit
under rocprofv3 (--kernel-trace --stats / --pmc).  usage: many_probe.py B N K [variant] [launches]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uamenv  # noqa: E402

B, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = sys.argv[4] if len(sys.argv) > 4 else "continuous"
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 20
env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=0, rng=os.environ.get("PROBE_RNG", "philox"),
                                      precision=os.environ.get("PROBE_PRECISION", "f32"))
env.reset()
dev = env.device
am = env.cfg.action_mode
if am == 2:
    tape = torch.rand((K, B), device=dev) * 2 - 1
elif am == 1:
    tape = torch.randint(0, 3, (K, B, 2), dtype=torch.int32, device=dev)
else:
    tape = torch.randint(0, 9 if am == 0 else 3, (K, B), dtype=torch.int32, device=dev)
rew = torch.empty((K, B), dtype=torch.float32, device=dev)
done = torch.empty((K, B), dtype=torch.uint8, device=dev)
info = torch.empty((K, B), dtype=torch.uint8, device=dev)
for _ in range(3):
    env.rollout_into(tape, rew, done, info)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(launches):
    env.rollout_into(tape, rew, done, info)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("B=%d N=%d K=%d %s: steps/launch %d, %.3f us/step, %.3f G env-steps/s" %
      (B, N, K, variant, env.steps_per_launch(K), el / (launches * K) * 1e6, B * launches * K / el / 1e9))
env.close()
