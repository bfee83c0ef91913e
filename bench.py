#!/usr/bin/env python3
"""Benchmark of the hot path: batched SingleAircraftEnv.step() on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is ONE fused uam_step launch over the whole per-GPU batch (BASELINE.json configs[2]:
discrete-9, 65536 envs x 32 intruders, the configuration the metric is quoted on).  Inputs (env
state, action tape) are resident in HBM before the timed region; observations, rewards and dones
stay on the device.  With N>1 every rank owns 65536 envs (weak scaling, env_id_offset = rank*B so
the union is one consistent batch) and the per-chunk reward/done slabs are all-gathered with RCCL.

One step is one launch of uam_sweep_kernel: the fp64 ownship phase (one thread per env, handed over through
LDS) and the intruder sweep / reduction / respawn / obs pack (one lane group per env).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : for the dominant kernel (uam_sweep_kernel): algorithmic HBM bytes per launch
                 (SURVEY §8d accounting) / its average launch duration, measured live
                 with hipEvents bracketing that kernel on the launch stream in a separate 256-step pass
                 after the timed region, against the 8 TB/s HBM3E peak.  `traffic` is the PMC figure
                 (FETCH_SIZE x2 + WRITE_SIZE per launch) committed under profiles/, when present,
  cpu_baseline : the CPU oracle (a port of the reference's step()) timed on this host's cores on a
                 bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
B_PER_GPU = 65536
N_INTRUDERS = 32
TAPE_LEN = 64
GATHER_EVERY = 32


def cpu_baseline(nthreads, seconds_target=12.0):
    """Oracle (kind 'port') on a bounded sample: same env config, random actions, all host threads."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    B, N = 8192, N_INTRUDERS
    cfg = oracle.default_config(B, N, rng_mode=oracle.RNG_PHILOX, precision=oracle.POS_F32, auto_reset=1, seed=0)
    env = oracle.OracleBatch(cfg, oracle.MATH_PORTABLE, threads=nthreads)
    env.reset(want_obs=False)
    rs = np.random.RandomState(1)
    acts = rs.randint(0, 9, size=(TAPE_LEN, B)).astype(np.int32)
    for k in range(3):
        env.step(acts[k], want_obs=True)
    steps, t0 = 0, time.perf_counter()
    while True:
        for k in range(10):
            env.step(acts[(steps + k) % TAPE_LEN], want_obs=True)
        steps += 10
        el = time.perf_counter() - t0
        if el >= seconds_target or steps >= 2000:
            break
    env.close()
    return {"value": B * steps / el, "unit": "env-steps/s", "cores": nthreads, "kind": "port",
            "sample": "oracle/uam_oracle.c (OpenMP over envs), %d envs x %d intruders, %d steps, random actions, "
                      "float64 obs written" % (B, N, steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="envs per GPU")
    ap.add_argument("--intruders", type=int, default=N_INTRUDERS)
    ap.add_argument("--policy", default="random", choices=["random", "goal"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: gloo lets 2 ranks share one GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--exchange", default="all_gather", choices=["all_gather", "gather", "none"],
                    help="N>1: how the per-chunk reward/done slab is exchanged (gather = to rank 0 only; none = exploration)")
    ap.add_argument("--variant", default="discrete9", help="exploration only: env variant (default = the BASELINE workload)")
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--precision", default="f32", choices=["f32", "exact"])
    args = ap.parse_args()

    import torch
    import uamenv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    B, N = args.batch, args.intruders
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=args.variant, rng=args.rng, precision=args.precision, seed=0,
                                          auto_reset=True, device=local_rank, env_id_offset=rank * B)
    dev = env.device
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    if env.cfg.action_mode == 0:
        tape = torch.randint(0, 9, (TAPE_LEN, B), dtype=torch.int32, device=dev, generator=gen)
    elif env.cfg.action_mode == 1:
        tape = torch.randint(0, 3, (TAPE_LEN, B, 2), dtype=torch.int32, device=dev, generator=gen)
    else:
        tape = torch.rand((TAPE_LEN, B), dtype=torch.float32, device=dev, generator=gen) * 2 - 1
    # Chunk slabs: rewards (float32) and dones (uint8) of GATHER_EVERY steps live in ONE byte buffer so that a
    # single collective moves both; two slabs alternate so the gather of chunk c overlaps the steps of chunk c+1.
    KB = GATHER_EVERY * B

    def make_slab():
        raw = torch.empty((KB * 5,), dtype=torch.uint8, device=dev)
        return raw, raw[:KB * 4].view(torch.float32).view(GATHER_EVERY, B), raw[KB * 4:].view(GATHER_EVERY, B)

    slabs = [make_slab(), make_slab()]
    pending = [None, None]
    need_all = world > 1 and (args.exchange == "all_gather" or (args.exchange == "gather" and rank == 0))
    gathered = [torch.empty((world * KB * 5,), dtype=torch.uint8, device=dev) for _ in range(2)] if need_all else None
    acts = env.policy_actions("random")

    def run(nsteps, offset=0):
        for k in range(nsteps):
            i = offset + k
            if args.policy == "goal":
                a = env.policy_actions("goal", out=acts)
            else:
                a = tape[i % TAPE_LEN]
            # rewards/dones land directly in row j of the chunk slab: identical per-step work at every N
            c, j = (i // GATHER_EVERY) & 1, i % GATHER_EVERY
            if j == 0 and pending[c] is not None:
                pending[c].wait()            # the stream (not the host) waits until this slab's last gather is done
                pending[c] = None
            raw, rew_slab, done_slab = slabs[c]
            env.step_into(a, rew_slab[j], done_slab[j])
            if world > 1 and j == GATHER_EVERY - 1 and args.exchange != "none":
                # the one exchange of the path: the chunk's reward/done slab over xGMI (RCCL)
                if args.exchange == "all_gather":
                    pending[c] = dist.all_gather_into_tensor(gathered[c], raw, async_op=True)
                else:
                    pending[c] = dist.gather(raw, gather_list=list(gathered[c].view(world, -1).unbind(0)) if rank == 0 else None,
                                             dst=0, async_op=True)

    def drain():
        for c in (0, 1):
            if pending[c] is not None:
                pending[c].wait()
                pending[c] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    drain()
    fence()
    env.timing_begin()
    t0 = time.perf_counter()
    run(args.steps, offset=args.warmup)
    drain()
    kernel_ms = env.timing_end()      # hipEvents on the launch stream, averaged over the uam_step launches
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations: a separate pass with hipEvents around each launch (not part of `value`)
    env.kernel_timing(True)
    run(256, offset=args.warmup + args.steps)
    drain()
    own_ms, sweep_ms, nsamp = env.kernel_timing_read()
    env.kernel_timing(False)

    if rank == 0:
        total_env_steps = float(B) * world * args.steps
        value = total_env_steps / elapsed
        step_bytes = env.step_bytes                      # (40 N + 150) B per env-step, SURVEY §8d
        sweep_bytes = step_bytes                         # the step is one launch: it moves all of the step's bytes
        # the hipEvents that bracket the timed region (divided by the launches) give the kernel's average duration
        # without the perturbation of per-launch event records (`bracketed_launch_ms` is that perturbed figure)
        dom_ms = kernel_ms
        achieved = sweep_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc_path) and B == B_PER_GPU and N == N_INTRUDERS:
            with open(pmc_path) as f:
                traffic = json.load(f).get("uam_sweep_kernel", {}).get("hbm_bytes_per_launch")
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("SingleAircraftEnv discrete-9, batch=%d envs x %d intruders per GPU "
                                    "(BASELINE.json configs[2], roofline run)" % (B, N)) if args.variant == "discrete9"
                       else "%s variant, batch=%d x %d intruders per GPU" % (args.variant, B, N),
                       "envs_per_gpu": B, "intruders": N, "policy": args.policy, "rng": args.rng,
                       "variant": args.variant, "positions": "float32" if args.precision == "f32" else "float64 (exact promotion)", "obs": "float32[B,%d] written every step" % env.obs_dim,
                       "exchange": "none" if world == 1 or args.exchange == "none" else
                                   "one async %s of the [%d, B] reward+done slab per chunk, double-buffered "
                                   "(overlaps the next chunk)" % (args.exchange, GATHER_EVERY)},
            "pair_steps_per_sec": value * N,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "uam_sweep_kernel (ownship phase + intruder sweep, one launch per step)",
                         "bytes_per_launch": sweep_bytes,
                         "avg_launch_ms": dom_ms, "bracketed_launch_ms": sweep_ms, "sampled_launches": nsamp,
                         "own_kernel_avg_ms": own_ms,
                         "whole_step": {"bytes": step_bytes, "avg_ms": kernel_ms,
                                        "frac": step_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(min(16, len(os.sched_getaffinity(0))))
            except Exception as ex:   # the checker must never take the measurement down
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (ex,)}
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
