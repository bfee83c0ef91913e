#!/usr/bin/env python3
"""Development tool: device time per step of any variant with uam_config overrides.
  python tools/bench_variants.py her9 shaped_reward=1 -- her3 -- discrete9"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, uamenv

def run(variant, over, B=65536, N=32, T=600):
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=0, **over)
    env.reset()
    for _ in range(400):      # past the start-up transient (first goal arrivals / collisions)
        env.step(env.policy_actions("random", 1))
    acts = [env.policy_actions("random", 1) for _ in range(8)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dones = torch.zeros((), dtype=torch.float64, device=env.device)
    ev0.record()
    for t in range(T):
        o, r, d, i = env.step(acts[t % 8])
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / T
    for t in range(100):
        o, r, d, i = env.step(acts[t % 8])
        dones += d.view(torch.uint8).sum()
    print(json.dumps({"variant": variant, "over": over, "us_per_step": round(us, 2), "done_rate": float(dones) / (100 * B),
                      "bytes": env.step_bytes, "GBps": round(env.step_bytes / us / 1e3, 1)}))
    env.close()

args, cur = sys.argv[1:], []
groups = []
for a in args + ["--"]:
    if a == "--":
        if cur: groups.append(cur)
        cur = []
    else:
        cur.append(a)
for g in groups:
    over = {}
    for kv in g[1:]:
        k, v = kv.split("=")
        over[k] = float(v) if "." in v else int(v)
    run(g[0], over)
