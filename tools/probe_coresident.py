#!/usr/bin/env python3
"""Development tool: how much does a co-resident kernel (stand-in for RCCL's) cost the single-round sweep grid?
Steps 65536x32 while a spin kernel of W workgroups sits on a side stream."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, uamenv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = uamenv.BatchedSingleAircraftEnv(B, 32, seed=0)
env.reset()
acts = [env.policy_actions("random", seed=k) for k in range(8)]
side = torch.cuda.Stream()
def run(T):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(T):
        env.step(acts[t % 8])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / T
run(200)
base = run(1000)
out = {"B": B, "alone_us": round(base, 2)}
streams = [torch.cuda.Stream() for _ in range(128)]
for W in (1, 8, 32, 128):
    # W spinning single-wave workgroups for ~80 ms, one per side stream (torch.cuda._sleep = 1 block, 1 thread)
    torch.cuda.synchronize()
    for st in streams[:W]:
        with torch.cuda.stream(st):
            torch.cuda._sleep(int(80e-3 * 2.1e9))
    time.sleep(0.002)
    t = run(1500)      # ~25 ms of stepping, inside the sleepers' lifetime
    torch.cuda.synchronize()
    out["with_%d_spinners_us" % W] = round(t, 2)
print(json.dumps(out))
