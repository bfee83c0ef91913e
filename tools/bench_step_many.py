#!/usr/bin/env python3
"""Development tool: per-step cost of env.step() from Python vs uam_step_many (K steps per C call)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, uamenv
for B, N in ((4096, 8), (65536, 32), (256, 8)):
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=0)
    env.reset()
    K = 64
    tape = torch.randint(0, 9, (K, B), dtype=torch.int32, device=env.device)
    for _ in range(3):
        env.step_many(tape, want_obs=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        env.step_many(tape)
    torch.cuda.synchronize()
    many = (time.perf_counter() - t0) / (20 * K)
    t0 = time.perf_counter()
    for r in range(20 * K):
        env.step(tape[r % K])
    torch.cuda.synchronize()
    single = (time.perf_counter() - t0) / (20 * K)
    print(json.dumps({"B": B, "N": N, "us_per_step_python_loop": single * 1e6, "us_per_step_step_many": many * 1e6,
                      "env_steps_per_s_step_many": B / many}))
    env.close()
