#!/usr/bin/env python3
"""Closed-loop stepping of small batches: a fresh action from a device-side policy before EVERY step (the goal-seeking
policy kernel stands in for a learner's forward pass).  Eager = one Python iteration per step (policy launch, step
launch(es)); graph = the same sequence captured once into a HIP graph of `--steps-per-graph` steps and replayed
(uam_config.graph_safe).  Prints one JSON line per (shape, mode).   usage: bench_closed_loop.py [--steps-per-graph 16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uamenv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps-per-graph", type=int, default=16)
ap.add_argument("--total", type=int, default=4096)
args = ap.parse_args()
for variant, B, N in (("discrete9", 4096, 8), ("randintruder", 8192, 64), ("continuous", 8192, 32), ("discrete9", 65536, 32)):
    for mode in ("eager", "graph", "in_kernel"):
        env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=0, graph_safe=int(mode == "graph"))
        env.reset()
        acts = env.policy_actions("goal")
        if mode == "in_kernel":
            # uam_rollout_policy: the policy runs inside the looped kernel (ownship wave), 64 steps per launch
            K = 64
            rew = torch.empty((K, B), dtype=torch.float32, device=env.device)
            done = torch.empty((K, B), dtype=torch.uint8, device=env.device)
            info = torch.empty((K, B), dtype=torch.uint8, device=env.device)
            acts_k = torch.empty((K, B, 2) if env.cfg.action_mode == 1 else (K, B), dtype=acts.dtype, device=env.device)

            def run(n):
                for _ in range(n // K):
                    env.rollout_policy(K, kind="goal", rew_out=rew, done_out=done, info_out=info, actions_out=acts_k)
        elif mode == "eager":
            def run(n):
                for _ in range(n):
                    env.policy_actions("goal", out=acts)
                    env.step(acts)
        else:
            loop = uamenv.GraphedLoop(env, lambda obs: env.policy_actions("goal", out=acts), steps=args.steps_per_graph)

            def run(n):
                for _ in range(n // args.steps_per_graph):
                    loop.replay()
        run(512)
        torch.cuda.synchronize()
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(args.total)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / args.total * 1e6)
        reps.sort()
        print(json.dumps({"variant": variant, "envs": B, "intruders": N, "mode": mode, "us_per_step": reps[1],
                          "env_steps_per_sec": B / (reps[1] * 1e-6), "policy": "goal-seeking kernel before every step",
                          "steps_per_graph": args.steps_per_graph if mode == "graph" else None,
                          "steps_per_launch": env.steps_per_launch(64) if mode == "in_kernel" else 1}))
        sys.stdout.flush()
        env.close()
