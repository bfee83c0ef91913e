#!/usr/bin/env python3
"""Development probe: does stepping S independent sub-batches on S streams beat one launch per step over the whole batch?
The step launches of one stream serialise (the tail of step k and the ramp of step k+1 never overlap); sub-batches on
their own streams are independent (envs never interact), so one's tail can overlap another's body.
usage: split_stream_probe.py B N K [S ...]   (fuse_rollout = never: one launch per step and sub-batch)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uamenv  # noqa: E402

B, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
splits = [int(s) for s in sys.argv[4:]] or [1, 2, 4]
variant = os.environ.get("PROBE_VARIANT", "discrete9")
dev = torch.device("cuda", 0)
reps = int(os.environ.get("PROBE_REPS", "12"))
same_stream = os.environ.get("PROBE_SAME_STREAM", "0") == "1"     # the sub-batches take turns on ONE stream


def run(S):
    Bs = B // S
    envs, streams, bufs = [], [], []
    for s in range(S):
        e = uamenv.BatchedSingleAircraftEnv(Bs, N, variant=variant, seed=0, env_id_offset=s * Bs, fuse_rollout=2)
        e.reset()
        envs.append(e)
        streams.append(torch.cuda.Stream(device=dev) if not (same_stream and s) else streams[0])
        tape = torch.randint(0, 9, (K, Bs), dtype=torch.int32, device=dev) if e.cfg.action_mode == 0 else \
            torch.rand((K, Bs), device=dev) * 2 - 1
        bufs.append((tape, torch.empty((K, Bs), dtype=torch.float32, device=dev),
                     torch.empty((K, Bs), dtype=torch.uint8, device=dev), torch.empty((K, Bs), dtype=torch.uint8, device=dev)))
    torch.cuda.synchronize()

    def chunk():
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                t, r, d, i = bufs[s]
                envs[s].rollout_into(t, r, d, i)

    for _ in range(3):
        chunk()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            chunk()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / (4 * K) * 1e6)
    times.sort()
    for e in envs:
        e.close()
    return {"streams": S, "envs_per_stream": Bs, "us_per_step_median": times[len(times) // 2], "us_per_step_min": times[0],
            "G_env_steps_per_s": B / times[len(times) // 2] / 1e3, "same_stream": same_stream}


for S in splits:
    print(json.dumps(dict(B=B, N=N, K=K, variant=variant, **run(S))))
    sys.stdout.flush()
