#!/usr/bin/env python3
"""Development benchmark of the batched MCTS planner (BASELINE.json configs[4]: RandIntruder env,
batch=8192 x 64 intruders, driven by MCTS): decisions/s on the GPU, the CPU oracle beside it."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import uamenv

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--intruders", type=int, default=64)
ap.add_argument("--sims", type=int, default=100)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cpu-roots", type=int, default=16)
ap.add_argument("--model", default="randintru", choices=["randintru", "straight4"],
                help="randintru: nodes_single_randintru.py on the RandIntruder env (Agent_RandInt.py); straight4: nodes_single.py on SingleAircraftMCTSEnv (Agent.py)")
args = ap.parse_args()

B, N = args.batch, args.intruders
env = uamenv.BatchedSingleAircraftEnv(B, N, variant="randintruder" if args.model == "randintru" else "mcts", seed=0)
env.reset()
for _ in range(20):
    env.step(env.policy_actions("goal"))
planner = uamenv.BatchedMCTS(N, seed=0, simulations=args.sims, search_depth=args.depth, model=args.model)
roots = planner.roots_from_env(env)
planner.best_action(roots)
torch.cuda.synchronize()
ts = []
for r in range(args.reps):
    t0 = time.perf_counter()
    acts = planner.best_action(roots, root_id0=r * B)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
gpu_s = sorted(ts)[len(ts) // 2]
# the Agent_RandInt.py loop: plan every 5th env step
t0 = time.perf_counter()
steps = 50
for t in range(steps):
    if t % 5 == 0:
        acts = planner.best_action(planner.roots_from_env(env), root_id0=t * B)
    env.step(acts)
torch.cuda.synchronize()
loop_s = time.perf_counter() - t0
out = {"batch": B, "intruders": N, "sims": args.sims, "depth": args.depth, "model": args.model,
       "gpu_ms_per_batched_decision": gpu_s * 1e3, "gpu_decisions_per_s": B / gpu_s,
       "agent_loop_env_steps_per_s": B * steps / loop_s}
try:
    from oracle import oracle
    ocfg = oracle.UamMctsConfig()
    for name, _ in oracle.UamMctsConfig._fields_:
        setattr(ocfg, name, getattr(planner.cfg, name))
    rs = roots[:args.cpu_roots].cpu().numpy()
    t0 = time.perf_counter()
    for k in range(len(rs)):
        oracle.mcts_best_action(ocfg, rs[k], rng_mode=oracle.RNG_PHILOX, math_impl=oracle.MATH_PORTABLE, root_id=k)
    cpu = (time.perf_counter() - t0) / len(rs)
    out["cpu_oracle_ms_per_decision_1core"] = cpu * 1e3
    out["speedup_vs_1core_oracle"] = (B / gpu_s) * cpu
except Exception as ex:
    out["cpu_oracle"] = repr(ex)
print(json.dumps(out))
