#!/usr/bin/env python3
"""Development tool: what a plain device copy achieves at the sweep kernel's footprint (context for roofline.frac).
Traffic = read + write bytes; sizes chosen to match one batched step at 65536x32 (94 MB) and at 1M x 32 (1.5 GB)."""
import json, torch
for total_mb in (94, 375, 1500):
    n = total_mb * 1000 * 1000 // 8          # float32 elements per side (read n*4 + write n*4)
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(20):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(json.dumps({"copy_traffic_MB": total_mb, "us": round(us, 2), "GBps": round(n * 8 / us / 1e3, 1)}))
    del a, b
