#!/usr/bin/env python3
"""Development tool: throughput of the hindsight-relabel kernels (rows/s and GB/s of algorithmic traffic)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, uamenv
M, D = 1 << 24, 26
x = torch.rand((M, D), dtype=torch.float32, device="cuda")
a, g = x[:, :2].contiguous(), x[:, -2:].contiguous()
for name, fn, nbytes in (("input_reward", lambda: uamenv.her_input_reward(x, num_intruders=8, knn=4), M * (D * 4 + 5)),
                         ("goal_reward", lambda: uamenv.her_goal_reward(a, g), M * 20)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"kernel": name, "rows": M, "ms": round(ms, 4), "Grows_per_s": round(M / ms / 1e6, 2), "GBps": round(nbytes / ms / 1e6, 1)}))
