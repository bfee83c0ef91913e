#!/usr/bin/env python3
"""Benchmark of the hot path: batched SingleAircraftEnv.step() on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path (one fused uam_sweep_kernel launch) over the rank's whole batch.  Inputs (env
state, action tape) are resident in HBM before the timed region; observations (written every step), rewards and
dones stay on the device.  Steps are issued in rollout chunks through the C-ABI (`uam_rollout`: the K launches of a
chunk are queued by one call, no Python between them), and with N>1 the chunk's reward/done slab is exchanged with
ONE asynchronous RCCL collective that overlaps the next chunk (uamenv.sharded.RolloutExchange).

N>1 without WORLD_SIZE in the environment: this process never touches the GPU; it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, passes its output through and exits
with its status.  Under torch.distributed.run (how the driver starts N>1) every rank runs main() directly.

Two workloads are measured in every invocation, the first is `value`:
  weak   (default)   BASELINE.json configs[2]: SingleAircraftEnv discrete-9, 65536 envs x 32 intruders PER GPU
                     (the configuration the metric is quoted on; at N=1 it is exactly the roofline run).
  strong (field)     BASELINE.json configs[3]: SingleAircraft2Env continuous, ONE batch of 65536 envs x 32
                     intruders partitioned over the ranks by shard_bounds (8192 per GPU at N=8); small shards run
                     their chunk as one looped launch (uam_sweep_many_kernel).
`--scaling strong` swaps them (`--variant` then picks the env of the primary workload).

Prints ONE JSON line (rank 0): the contract fields, plus
  reps_ms      : wall time of every repetition of the K-step region; `ms_per_step`/`value` come from the median
  roofline     : dominant kernel of the primary workload: algorithmic HBM bytes per launch (SURVEY §8d accounting,
                 DESIGN.md §5) / its average duration from hipEvents bracketing the timed region on the launch stream,
                 against the 8 TB/s HBM3E peak; `hbm_resident` repeats it at 1 M envs (1.5 GB per step, beyond the
                 256 MiB Infinity Cache) next to a plain device copy of the same bytes
  cpu_baseline : the CPU oracle (a C port of the reference's step()) on this host's cores, bounded sample (N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
GLOBAL_BATCH = 65536    # BASELINE.json configs[2]/[3]
N_INTRUDERS = 32
TAPE_LEN = 64
PMC_PROFILE = "profiles/r04_pmc_summary.json"
# measured in the build container by importing the reference; cannot be re-measured on the GPU box
REF_PYTHON = {"env_steps_per_sec_per_core": 3.4e3, "where": "reference NumPy step() at N=32, build container "
              "(8 vCPU Xeon 2.1 GHz), one process per core; BASELINE.md §2 — measured elsewhere, not on this box"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step region (median is reported)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="which workload is `value`")
    ap.add_argument("--batch", type=int, default=GLOBAL_BATCH, help="weak: envs per GPU; strong: envs in total")
    ap.add_argument("--intruders", type=int, default=N_INTRUDERS)
    ap.add_argument("--variant", default=None, help="env of the primary workload (default: discrete9 for weak, "
                    "continuous for strong = BASELINE configs[2] / [3])")
    ap.add_argument("--policy", default="random", choices=["random", "goal"],
                    help="random: action tape in HBM; goal: goal-seeking policy kernel before every step (closed loop)")
    ap.add_argument("--chunk", type=int, default=0, help="steps per rollout chunk / exchange (0 = 32 at 65536 envs "
                    "per GPU, growing to 256 for smaller shards)")
    ap.add_argument("--obs-slots", type=int, default=1, help="observation slots of a chunk: 1 = one [B,D] buffer "
                    "overwritten every step, 2 = double buffer, 0 = the full [K,B,D] rollout slab")
    ap.add_argument("--exchange", default="gather", choices=["all_gather", "gather", "none"],
                    help="N>1: the per-chunk collective of the reward/done slab: gather = to rank 0, the master that "
                         "SubprocVecEnv.step_wait collects in (north_star's single RCCL gather); all_gather = every rank")
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--precision", default="f32", choices=["f32", "exact"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="length of the closing sustained pass of the primary workload (0 = skip; profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second workload and the 1 M-env pass")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: gloo lets 2 ranks share one GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--dump-exchange", default=None, metavar="NPZ",
                    help="rehearsal only: after the measurements, step ONE partitioned batch for two chunks from a tape "
                         "every rank derives from the same seed and let rank 0 save what its collectives gathered "
                         "(tests compare it byte for byte with one unsharded handle)")
    ap.add_argument("--no-build", action="store_true", help="do not run make first (profiling runs: the profiled "
                    "process must not spawn children)")
    return ap.parse_args(argv)


def self_launch(args):
    """N>1 and no rendezvous in the environment: start the ranks as children.  Runs BEFORE anything imports torch
    or touches the GPU in this process (a process that has initialised the GPU must never exec/fork ranks)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # RCCL maps its peers' buffers through HIP IPC handles.  This image's host driver supports dmabuf IPC only: with the
    # legacy mode left on, hipIpcGetMemHandle fails with "invalid argument" and the communicator never comes up.  The
    # image exports the variable already; it is set here too so that a caller's scrubbed environment cannot lose it.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(B, N, nthreads, scale=1.0):
    """Oracle (kind 'port') on a bounded sample of the same workload: all host threads, then one thread."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    cfg = oracle.default_config(B, N, rng_mode=oracle.RNG_PHILOX, precision=oracle.POS_F32, auto_reset=1, seed=0)
    rs = np.random.RandomState(1)
    acts = rs.randint(0, 9, size=(8, B)).astype(np.int32)

    def rate(threads, seconds):
        env = oracle.OracleBatch(cfg, oracle.MATH_PORTABLE, threads=threads)
        env.reset(want_obs=False)
        env.step(acts[0], want_obs=True)
        steps, t0 = 0, time.perf_counter()
        while True:
            env.step(acts[steps % 8], want_obs=True)
            steps += 1
            el = time.perf_counter() - t0
            if el >= seconds or steps >= 4000:
                break
        env.close()
        return B * steps / el, steps

    aff = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    cores = max(1, min(aff, int(quota + 0.5))) if quota else aff      # the CPUs this process can actually keep busy
    # thread counts tried: the CPUs the cgroup grants, the pool's per-GPU share (16) and every CPU of the affinity mask;
    # `value` is the best of them — on a box whose CPU share is enforced by throttling, one thread per visible CPU is
    # far slower than one per granted CPU (measured: 256 threads 0.48 M, 16 threads 9.8 M env-steps/s)
    tried = {}
    for nt, secs in sorted({int(cores): 5.0, int(nthreads): 5.0, int(aff): 3.0}.items()):
        tried[nt] = rate(int(nt), secs * scale)
    best = max(tried, key=lambda k: tried[k][0])
    v_one, s_one = rate(1, 4.0 * scale)
    out = {"value": tried[best][0], "unit": "env-steps/s", "cores": best, "kind": "port",
           "sample": "oracle/uam_oracle.c (OpenMP over envs), %d envs x %d intruders, random actions, float64 obs written; "
                     "threads tried: %s (steps run: %s) — `value` is the best; affinity mask %d CPUs, cgroup quota %s; "
                     "+ %d steps on 1 thread; run BEFORE the GPU phases"
                     % (B, N, sorted(tried), [tried[k][1] for k in sorted(tried)], aff,
                        ("%.1f CPUs" % quota) if quota else "not visible", s_one),
           "by_threads": {str(k): tried[k][0] for k in sorted(tried)}, "single_thread": v_one, "nproc": os.cpu_count(),
           "affinity": aff, "cgroup_cpu_quota": quota, "reference_python": REF_PYTHON}
    return out


def cpu_quota():
    """CPUs of run time the cgroup grants this process (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            return float(q) / float(per)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return q / per if q > 0 else None
    except (OSError, ValueError):
        return None


class Ctx:
    pass


def median(xs):
    s = sorted(xs)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def measure(ctx, args, scaling, variant, steps, warmup, reps, want_bracketed, fuse="auto", sustain_s=0.0):
    """One workload on this rank's shard: spin-up, `warmup` untimed steps, then `reps` repetitions of exactly `steps`
    steps, each bracketed by barrier + synchronize; max over ranks per repetition."""
    torch, dist = ctx.torch, ctx.dist
    import uamenv
    from uamenv.sharded import RolloutExchange, shard_bounds
    from uamenv import _capi as capi
    N = args.intruders
    if scaling == "weak":
        counts = [args.batch] * ctx.world
        first = ctx.rank * args.batch
    else:
        spans = [shard_bounds(args.batch, ctx.world, r) for r in range(ctx.world)]
        counts = [c for _, c in spans]
        first = spans[ctx.rank][0]
    B = counts[ctx.rank]
    K = args.chunk if args.chunk > 0 else max(32, min(256, 32 * GLOBAL_BATCH // max(B, 1)))
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, rng=args.rng, precision=args.precision, seed=0,
                                          auto_reset=True, device=ctx.local_rank, env_id_offset=first, fuse_rollout=fuse)
    dev = env.device
    env.reset()
    if args.chunk <= 0 and int(env.steps_per_launch(K)) == 1 and ctx.world > 1 and steps < 4 * K:
        # a short region of single-step launches: four chunks (and four small slabs) instead of one, so that only the last
        # quarter's exchange is exposed at the closing barrier; chunking costs nothing here, the launches are the same
        K = max(4, -(-steps // 4))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + ctx.rank)
    am = env.cfg.action_mode
    TL = max(TAPE_LEN, K)
    if am == capi.ACT_DISCRETE9:
        tape = torch.randint(0, 9, (TL, B), dtype=torch.int32, device=dev, generator=gen)
    elif am == capi.ACT_PAIR:
        tape = torch.randint(0, 3, (TL, B, 2), dtype=torch.int32, device=dev, generator=gen)
    elif am == capi.ACT_DISCRETE3:
        tape = torch.randint(0, 3, (TL, B), dtype=torch.int32, device=dev, generator=gen)
    elif am == capi.ACT_CONTINUOUS2:
        tape = torch.rand((TL, B, 2), dtype=torch.float32, device=dev, generator=gen) * 2 - 1
    else:
        tape = torch.rand((TL, B), dtype=torch.float32, device=dev, generator=gen) * 2 - 1
    tape = torch.cat([tape, tape[:K]]).contiguous()       # any chunk of <= K rows is one contiguous slice
    R = K if args.obs_slots == 0 else args.obs_slots
    obs_slots = torch.empty((R, B, env.obs_dim), dtype=torch.float32, device=dev)
    info = torch.empty((K, B), dtype=torch.uint8, device=dev)
    ex = RolloutExchange(dist, K, counts, ctx.rank, dev, mode=args.exchange)
    looped = int(env.steps_per_launch(K)) > 1 and args.policy == "random"
    kernel_name = "%s (one launch per %s)" % (env.rollout_kernel(K if looped else 1), "chunk" if looped else "step")
    acts = env.policy_actions("random") if args.policy == "goal" else None
    state = {"chunk": 0, "launches": 0}
    bound = {}

    def run(nsteps, start):
        i = 0
        while i < nsteps:
            k = min(K, nsteps - i)
            c = state["chunk"] & 1
            ex.begin(c)                 # the stream (not the host) waits until this slab's last collective has read it
            rew, done = ex.slab(c)
            if args.policy == "goal":   # closed loop: policy kernel, then one step, per step
                for j in range(k):
                    env.policy_actions("goal", out=acts)
                    env.step_into(acts, rew[j], done[j])
            else:
                t0 = (start + i) % TL
                key = (t0, k, c)
                call = bound.get(key)
                if call is None:      # argument checks and pointer extraction once per distinct chunk (warm-up fills this)
                    call = bound[key] = env.bind_rollout(tape[t0:t0 + k], rew[:k], done[:k], info[:k], obs_slots)
                call()
            state["launches"] += 1 if (looped and k > 1) else k
            ex.exchange(c)              # the one exchange of the path: the chunk's reward/done slab over xGMI
            state["chunk"] += 1
            i += k

    spin_ev = torch.cuda.Event()

    def fence():
        # poll an event before the blocking synchronize: hipDeviceSynchronize sleeps on an interrupt and wakes tens of
        # microseconds after the queue drains, which is a tenth of a 20-step region
        spin_ev.record()
        while not spin_ev.query():
            pass
        torch.cuda.synchronize()
        if ctx.world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed spin-up (clock ramp, first-launch costs), not part of the driver-counted warmup.  Time-based, so it
    # must not contain a collective (ranks would disagree on the count): plain rollouts into slab 0, no exchange
    if args.policy == "random":
        t_spin = time.perf_counter()
        rew0, done0 = ex.slab(0)
        while time.perf_counter() - t_spin < 0.3:
            for _ in range(4):
                env.rollout_into(tape[:K], rew0, done0, info, obs_slots)
            torch.cuda.synchronize()
    run(warmup, 0)
    ex.drain()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_region(nsteps, start):
        """exactly `nsteps` steps bracketed by barrier + synchronize on both sides; max over ranks"""
        fence()
        state["launches"] = 0
        t0 = time.perf_counter()
        ev0.record()
        run(nsteps, start)
        ev1.record()
        ex.drain()
        fence()
        el = time.perf_counter() - t0
        if ctx.world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el * 1e3, ev0.elapsed_time(ev1), state["launches"]      # hipEvents on the launch stream around the launches

    # every repetition replays the same action rows into the same slab ring, so that the untimed rehearsal below has
    # already built every bound call the timed repetitions make
    timed_region(steps, warmup)
    timed_region(steps, warmup)
    reps_ms, span_ms, launches = [], [], []
    for r in range(reps):
        ms, sp, ln = timed_region(steps, warmup)
        reps_ms.append(ms); span_ms.append(sp); launches.append(ln)
    steady = None
    if steps < 1024 and args.policy == "random":
        # a short region (the driver times 20 steps) is mostly its fixed costs — the closing barrier, one exposed
        # exchange, the launch ramp; the same protocol over a long region says what the path sustains
        n_long = 2048
        timed_region(n_long, warmup)
        ms, _, _ = timed_region(n_long, warmup)
        steady = {"steps": n_long, "ms_per_step": ms / n_long, "value": float(sum(counts)) * n_long / (ms * 1e-3)}
    sustained = None
    if sustain_s > 0 and args.policy == "random":
        # seconds, not milliseconds, of the same protocol: the rate the path holds once clocks and temperatures have
        # settled (and a GPU phase long enough for an external utilisation sampler to see)
        per_step_ms = (steady["ms_per_step"] if steady else median(reps_ms) / steps)
        n_sus = max(K, int(sustain_s * 1e3 / per_step_ms) // K * K)
        ms, _, _ = timed_region(n_sus, warmup)
        sustained = {"seconds": ms * 1e-3, "steps": n_sus, "ms_per_step": ms / n_sus,
                     "value": float(sum(counts)) * n_sus / (ms * 1e-3)}
    out = {"scaling": scaling, "variant": variant, "envs_per_gpu": B, "envs_total": sum(counts), "chunk": K, "sustained": sustained,
           "obs_slots": R, "reps_ms": reps_ms, "median_ms": median(reps_ms),
           "value": float(sum(counts)) * steps / (median(reps_ms) * 1e-3),
           "ms_per_step": median(reps_ms) / steps,
           "steps_per_launch": (steps / launches[0]) if launches[0] else 0.0,
           "kernel": kernel_name,
           "stream_span_ms_per_step": median(span_ms) / steps, "launches": launches[0],
           "step_bytes": env.step_bytes, "obs_dim": env.obs_dim, "collectives_per_rank": ex.collectives,
           # algorithmic bytes of one launch: a step's bytes for the step kernel; for the looped kernel (state in
           # registers for the whole launch) the state once + K x the step's outputs and inputs
           "launch_bytes": env.rollout_bytes(max(1, int(round(steps / launches[0])))) if looped and launches[0] else env.step_bytes,
           "fuse": fuse, "steady_state": steady}
    if want_bracketed and not looped and args.policy == "random":
        # per-launch durations: a separate pass with hipEvents around each launch (serialises the queue slightly)
        env.kernel_timing(True)
        run(256, 0)
        ex.drain()
        _, sweep_ms, nsamp = env.kernel_timing_read()
        env.kernel_timing(False)
        out["bracketed_launch_ms"], out["sampled_launches"] = sweep_ms, nsamp
    ex.drain()
    torch.cuda.synchronize()
    env.close()
    del obs_slots, tape, info, ex
    return out


def config5_planner(ctx, reps=5):
    """BASELINE.json configs[4] beside the headline: the batched MCTS planner (8192 roots x 64 intruders, 100
    simulations, depth 3) that drives the RandIntruder env — ms per batched plan on this GPU, and the env steps per
    second of the reference's agent loop (Agent_RandInt.py: a plan every 5th step).  An extra: never `value`."""
    import time
    import uamenv
    torch = ctx.torch
    B, N = 8192, 64
    out = {}
    for key, variant, model, what in (
            ("agent_randint", "randintruder", "randintru", "RandIntruder env + nodes_single_randintru.py (Algorithms/MCTS/Agent_RandInt.py)"),
            ("agent", "mcts", "straight4", "SingleAircraftMCTSEnv + nodes_single.py, upstream quirks included (Algorithms/MCTS/Agent.py)")):
        out[key] = _planner_run(torch, uamenv, B, N, variant, model, what, reps)
    out["workload"] = ("BASELINE.json configs[4]: batch 8192 x 64 intruders driven by the batched MCTS planner (100 simulations, "
                       "depth 3, 10 frames per move), one wavefront per root; both of the reference's agent pairings")
    return out


def _planner_run(torch, uamenv, B, N, variant, model, what, reps):
    import time
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=0)
    env.reset()
    for _ in range(20):
        env.step(env.policy_actions("goal"))
    planner = uamenv.BatchedMCTS(N, seed=0, model=model)
    roots = planner.roots_from_env(env)
    planner.best_action(roots)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        acts = planner.best_action(roots, root_id0=(r + 1) * B)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    steps = 25
    t0 = time.perf_counter()
    for t in range(steps):
        if t % 5 == 0:
            acts = planner.best_action(planner.roots_from_env(env), root_id0=(reps + 2 + t) * B)
        env.step(acts)
    torch.cuda.synchronize()
    loop_s = time.perf_counter() - t0
    env.close()
    return {"pairing": what,
            "ms_per_batched_plan": ms[len(ms) // 2], "reps_ms": ms, "decisions_per_sec": B / (ms[len(ms) // 2] * 1e-3),
            "agent_loop_env_steps_per_sec": B * steps / loop_s,
            "agent_loop": "a plan every 5th env step (the loop of Agent.py / Agent_RandInt.py), %d steps, host-timed" % steps}


def goal_policy_extra(ctx, args, variant):
    """SURVEY 8d asks for both action regimes.  `value` is the random-action tape (episodes almost never end: the
    respawn-light steady state); this is the closed loop with the library's device-side goal-seeking heuristic (turn
    toward the goal, full throttle) choosing a FRESH action before every step from the state the previous step left —
    a policy launch and a step launch per step, queued by uam_rollout_policy, no Python between them — which brings
    goals, NMACs, conflicts and auto-resets (32 spawns each) into the timed steps.  An extra: never `value`."""
    torch = ctx.torch
    import uamenv
    B, N, K = args.batch, args.intruders, 32
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, rng=args.rng, precision=args.precision, seed=0,
                                          auto_reset=True, device=ctx.local_rank, fuse_rollout="never")
    dev = env.device
    env.reset()
    obs = torch.empty((1, B, env.obs_dim), dtype=torch.float32, device=dev)
    rew = torch.empty((K, B), dtype=torch.float32, device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev)
    info = torch.empty((K, B), dtype=torch.uint8, device=dev)
    two = env.cfg.action_mode in (1, 4)
    acts = torch.empty((K, B, 2) if two else (K, B), dtype=torch.float32 if env.cfg.action_mode in (2, 4) else torch.int32, device=dev)

    def chunk():
        env.rollout_policy(K, kind="goal", seed=3, rew_out=rew, done_out=done, info_out=info, obs_out=obs, actions_out=acts)

    for _ in range(10):        # 320 steps: goals start to be reached after ~150
        chunk()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times, ends = [], 0
    for _ in range(5):
        e0.record()
        for _ in range(8):
            chunk()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / (8 * K))
        ends += int(done.sum().item())
    ms = median(times)
    launches = int(env.policy_steps_per_launch(K))
    env.close()
    return {"policy": "goal-seeking heuristic on the device, a fresh action before every step (uam_rollout_policy)",
            "value": B / (ms * 1e-3), "ms_per_step": ms, "reps_ms_per_step": times,
            "launches_per_step": "one policy launch + one step launch" if launches == 1 else "%d steps per launch" % launches,
            "episode_ends_per_env_step": ends / float(5 * K * B),
            "roofline_frac": env_step_bytes(N) * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def env_step_bytes(N):
    return 40 * N + 150


def copy_ms(torch, dev, nbytes, iters):
    """Plain device copies that move `nbytes` in total (half read, half written): the practical HBM ceiling.  Two
    kernels: the library's own hand-written copy (uam_bench_copy: one 16-byte element per thread, the float4 copy
    MI355X_MICROARCH.md measures at 6.29 TB/s) and torch's copy kernel.  Returns (ms own, ms torch)."""
    from uamenv import _capi as capi
    n = (nbytes // 2) // 16 * 16
    src = torch.empty((n,), dtype=torch.uint8, device=dev)
    dst = torch.empty((n,), dtype=torch.uint8, device=dev)
    src.zero_()
    L = capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for own in (True, False):
        def once():
            if own:
                capi.check(L.uam_bench_copy(dev.index or 0, src.data_ptr(), dst.data_ptr(), n, st))
            else:
                dst.copy_(src)
        for _ in range(3):
            once()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            once()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters)
    return out[0], out[1]


def walk_gbs(torch, dev, stream_bytes, iters):
    """GB/s of the library's bare six-stream walk (uam_bench_walk: 4 reads, 2 in-place writes) at `stream_bytes` per stream."""
    from uamenv import _capi as capi
    stream_bytes = stream_bytes // 16 * 16
    spacing = stream_bytes + (2 << 20)          # separate hipMallocs of the arrays land size + 2 MiB apart
    buf = torch.empty((3 * spacing + stream_bytes,), dtype=torch.uint8, device=dev)
    buf.zero_()
    L = capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        capi.check(L.uam_bench_walk(dev.index or 0, buf.data_ptr(), 4, 2, stream_bytes, spacing, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        capi.check(L.uam_bench_walk(dev.index or 0, buf.data_ptr(), 4, 2, stream_bytes, spacing, st))
    e1.record()
    torch.cuda.synchronize()
    return 6.0 * stream_bytes / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9


def hbm_resident(ctx, args, variant):
    """The same kernel at 1 M envs: 1.5 GB of algorithmic traffic per step, beyond the 256 MiB Infinity Cache."""
    torch = ctx.torch
    import uamenv
    B, N = 1 << 20, args.intruders
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, rng=args.rng, precision=args.precision, seed=0,
                                          auto_reset=True, device=ctx.local_rank, fuse_rollout="never")   # the step kernel, one launch per step
    dev = env.device
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    am = env.cfg.action_mode
    if am in (2, 4):      # ACT_CONTINUOUS1 / ACT_CONTINUOUS2 (float32, one or two words per env)
        tape = torch.rand((8, B, 2) if am == 4 else (8, B), dtype=torch.float32, device=dev, generator=gen) * 2 - 1
    else:
        tape = torch.randint(0, 9 if am == 0 else 3, (8, B, 2) if am == 1 else (8, B), dtype=torch.int32, device=dev,
                             generator=gen)
    rew = torch.empty((8, B), dtype=torch.float32, device=dev)
    done = torch.empty((8, B), dtype=torch.uint8, device=dev)
    info = torch.empty((8, B), dtype=torch.uint8, device=dev)
    for _ in range(12):
        env.rollout_into(tape, rew, done, info)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(5):
        e0.record()
        for _ in range(8):
            env.rollout_into(tape, rew, done, info)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 64)
    ms = median(times)
    nbytes = env.step_bytes
    # the kernel alone: hipEvents around EACH launch (serialised, ~2 % event overhead at this size).  The span above also
    # holds the gaps between launches and the noise launch that is drawn ahead on a side stream WHILE steps run (at
    # 1 M envs a ~320 us kernel per 16 steps competing for the same CUs); rocprofv3's per-kernel average is this figure
    kms = None
    try:
        env.kernel_timing(True)
        for _ in range(8):
            env.rollout_into(tape, rew, done, info)
        torch.cuda.synchronize()
        _, kms, nk = env.kernel_timing_read()
        env.kernel_timing(False)
        if not nk:
            kms = None
    except Exception:
        kms = None
    env.close()
    del tape, rew, done, info
    torch.cuda.empty_cache()
    cms, cms_torch = copy_ms(torch, dev, nbytes, 10)
    ach = nbytes / (ms * 1e-3) / 1e9
    walk = None
    try:
        walk = walk_gbs(torch, dev, B * N * 4, 10)
    except Exception:       # an extra of an extra
        walk = None
    out = {"envs": B, "bytes_per_launch": nbytes, "avg_launch_ms": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
           "copy_same_bytes_ms": cms, "frac_of_copy": cms / ms, "copy_kernel": "uam_copy_flat_kernel (library: one 16-byte element per thread, grid as large as the buffer)",
           "copy_frac_of_peak": nbytes / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, "torch_copy_same_bytes_ms": cms_torch,
           "launches_timed": 320, "reps_ms": times,
           "note": "frac: hipEvents around 5 x 64 queued launches (stream span / launches: includes launch gaps and the "
                   "overlapped noise launches), median; kernel_frac: hipEvents around each of 64 launches (the step "
                   "kernel alone, what rocprofv3 reports); working set 1.1 GB state+obs, not cache-resident"}
    if walk:
        out["walk_six_streams"] = {"achieved": walk, "frac_of_peak": walk / HBM_PEAK_GBS, "step_frac_of_walk": ach / walk,
                                   "kernel": "uam_walk_kernel (library): four read streams of the per-intruder arrays' size, two of "
                                             "them overwritten in place, one 16-byte element per thread — the step's fat streams "
                                             "(px py vx vy in, px py out) with no arithmetic; arrays spaced like the handle's",
                                   "why": "a kernel with six streams does not reach the two-stream copy's rate on this memory system "
                                          "(profiles/r04_streams_and_placement.md)"}
    if kms:
        out["kernel_launch_ms"] = kms
        out["kernel_achieved"] = nbytes / (kms * 1e-3) / 1e9
        out["kernel_frac"] = out["kernel_achieved"] / HBM_PEAK_GBS
    return out


def exchange_dump(ctx, args, path):
    """The exchange path with known inputs: the strong partition of ONE continuous batch, two chunks of 16 steps from a
    tape that is a function of the global env index only, each chunk's reward/done slab through the rank's
    RolloutExchange exactly as measure() drives it.  Rank 0 saves the gathered [2 * 16, B_total] rewards and dones."""
    import numpy as np
    torch, dist = ctx.torch, ctx.dist
    import uamenv
    from uamenv.sharded import RolloutExchange, shard_bounds
    Btot, N, K = args.batch, args.intruders, 16
    spans = [shard_bounds(Btot, ctx.world, r) for r in range(ctx.world)]
    first, B = spans[ctx.rank]
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant="continuous", seed=0, auto_reset=True, device=ctx.local_rank,
                                          env_id_offset=first)
    dev = env.device
    env.reset()
    tape_all = np.random.RandomState(1234).uniform(-1.0, 1.0, size=(2 * K, Btot)).astype(np.float32)
    tape = torch.from_numpy(np.ascontiguousarray(tape_all[:, first:first + B])).to(dev)
    info = torch.empty((K, B), dtype=torch.uint8, device=dev)
    mode = args.exchange if args.exchange != "none" else "gather"
    ex = RolloutExchange(dist, K, [c for _, c in spans], ctx.rank, dev, mode=mode, force=dist is not None)
    for c in (0, 1):
        ex.begin(c)
        rew, done = ex.slab(c)
        env.rollout_into(tape[c * K:(c + 1) * K].contiguous(), rew, done, info)
        ex.exchange(c)
    ex.drain()
    torch.cuda.synchronize()
    if ctx.rank == 0:
        r0, d0 = ex.gathered(0)
        r1, d1 = ex.gathered(1)
        np.savez(path, rew=torch.cat([r0, r1]).cpu().numpy(), done=torch.cat([d0, d1]).cpu().numpy(), tape=tape_all,
                 intruders=N, world=ctx.world, mode=mode, collectives=ex.collectives,
                 steps_per_launch=int(env.steps_per_launch(K)))
    env.close()


def ensure_built():
    """The HIP library normally travels prebuilt; if it is missing or stale and a toolchain is here, build it ONCE, in
    the process that starts the ranks (never in the ranks: N concurrent makes in one directory).  No torch, no GPU."""
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gym-guidance-collision-avoidance-single_amd", "csrc")
    if os.environ.get("UAMENV_LIB"):
        return
    try:
        r = subprocess.run(["make", "-C", csrc, "-s", "-j2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=False)
        if r.returncode != 0:       # never benchmark a stale library in silence
            sys.stderr.write("bench.py: `make -C %s` failed (rc %d); the library on disk, if any, is STALE:\n%s\n"
                             % (csrc, r.returncode, r.stdout.decode("utf-8", "replace")[-2000:]))
    except OSError:
        pass            # no make here: uamenv raises a clear ImportError if the library really is missing


def under_profiler():
    """True when a profiler's preloaded library shares this process (rocprofv3 exports ROCPROF* and preloads
    librocprofiler-sdk-tool.so): with --pmc it initialises the GPU before Python starts, and a GPU-initialised process
    must not spawn children here (ensure_built's `make`, the CPU baseline's worker threads are fine, a fork+exec is not)."""
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)


def main():
    args = parse_args()
    if under_profiler():
        # a profiled run measures kernels, not the job: no child processes, no seconds-long closing pass
        args.no_build = True
        args.sustain_seconds = 0.0
    if "WORLD_SIZE" not in os.environ and not args.no_build:
        ensure_built()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        # FIRST, before this process touches the GPU: the CPU oracle on the host cores (the GPU phases then close the run)
        try:
            cpu = cpu_baseline(args.batch, args.intruders, min(16, len(os.sched_getaffinity(0))))
        except Exception as ex:   # the checker must never take the measurement down
            cpu = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}

    import torch
    ctx = Ctx()
    ctx.torch = torch
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = int(os.environ.get("RANK", "0"))
    ctx.local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if ctx.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, ctx.world))
    torch.cuda.set_device(ctx.local_rank)
    ctx.dist = None
    if ctx.world > 1:
        import torch.distributed as dist
        ctx.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", ctx.local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    primary = args.scaling
    secondary = "strong" if primary == "weak" else "weak"
    default_variant = {"weak": "discrete9", "strong": "continuous"}
    pvar = args.variant or default_variant[primary]
    # `value` of the weak workload is ONE LAUNCH PER STEP in every mode (the closed-loop-capable path: a policy can act
    # between any two steps), so its chunks are never fused; the shards of the strong workload run in whatever form the
    # library picks for their size (fuse_rollout = auto: one launch per chunk); `fused_rollout` reports the weak
    # workload with fused chunks
    fuse_of = {"weak": "never", "strong": "auto"}
    res = measure(ctx, args, primary, pvar, args.steps, args.warmup, max(1, args.reps), want_bracketed=True, fuse=fuse_of[primary])
    sec = None
    if not args.no_secondary:
        sec = measure(ctx, args, secondary, default_variant[secondary], args.steps, args.warmup, max(1, min(3, args.reps)),
                      want_bracketed=False, fuse=fuse_of[secondary])
    fused = None
    if ctx.world == 1 and not args.no_secondary and args.policy == "random":
        # the same workload with the chunk FUSED into one looped launch per chunk (fuse_rollout = always): open-loop
        # only, state register-resident for the whole chunk — reported beside `value`, never as `value`
        fused = measure(ctx, args, primary, pvar, args.steps, args.warmup, max(1, min(3, args.reps)), want_bracketed=False,
                        fuse="always")
    resident = None
    if ctx.world == 1 and not args.no_secondary and args.policy == "random":
        try:
            resident = hbm_resident(ctx, args, pvar)
        except Exception as ex:       # an extra; must never take the measurement down
            resident = {"error": repr(ex)}
    goal = None
    if ctx.world == 1 and not args.no_secondary and args.policy == "random":
        try:
            goal = goal_policy_extra(ctx, args, pvar)
        except Exception as ex:       # an extra; must never take the measurement down
            goal = {"error": repr(ex)}
    planner = None
    if ctx.world == 1 and not args.no_secondary and args.policy == "random" and args.rng == "philox":
        try:
            planner = config5_planner(ctx)
        except Exception as ex:       # an extra; must never take the measurement down
            planner = {"error": repr(ex)}
    sustained = None
    if args.sustain_seconds > 0 and args.policy == "random":
        # the closing GPU phase: the primary workload for seconds (the CPU baseline ran first, so the device is busy at
        # the end of the run, not idle behind a CPU loop)
        sustained = measure(ctx, args, primary, pvar, args.steps, args.warmup, 1, want_bracketed=False,
                            fuse=fuse_of[primary], sustain_s=args.sustain_seconds)["sustained"]

    if ctx.rank == 0:
        N = args.intruders
        spl = max(1.0, res["steps_per_launch"])
        bytes_per_launch = float(res["launch_bytes"])
        launch_ms = res["stream_span_ms_per_step"] * spl
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, PMC_PROFILE)
        if os.path.exists(pmc_path) and res["envs_per_gpu"] == GLOBAL_BATCH and N == N_INTRUDERS and pvar == "discrete9":
            with open(pmc_path) as f:
                traffic = json.load(f).get("uam_sweep_kernel", {}).get("hbm_bytes_per_launch")
            traffic_src = PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE in its own run; not measured in this invocation)"
        ccopy, ccopy_torch = copy_ms(torch, torch.device("cuda", ctx.local_rank), int(bytes_per_launch), 50) if ctx.world == 1 else (None, None)
        workloads = {"weak": "SingleAircraftEnv discrete-9, batch=%d envs x %d intruders per GPU (BASELINE.json configs[2], "
                             "roofline run)" % (args.batch, N),
                     "strong": "SingleAircraft2Env continuous, ONE batch of %d envs x %d intruders partitioned over the "
                               "GPUs (BASELINE.json configs[3])" % (args.batch, N)}
        wl = workloads[primary] if pvar == default_variant[primary] else \
            "%s variant, %s scaling, batch=%d x %d intruders" % (pvar, primary, args.batch, N)
        out = {
            "metric": "env_steps_per_sec", "value": res["value"], "unit": "env-steps/s", "n_gpus": ctx.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": primary, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "envs_per_gpu": res["envs_per_gpu"], "envs_total": res["envs_total"],
                       "intruders": N, "policy": args.policy, "rng": args.rng, "variant": pvar,
                       "positions": "float32" if args.precision == "f32" else "float64 (exact promotion)",
                       "obs": "float32[B,%d] written every step (%d slot%s)" % (res["obs_dim"], res["obs_slots"],
                                                                                "" if res["obs_slots"] == 1 else "s"),
                       "chunk": res["chunk"], "launch": res["kernel"],
                       "exchange": "none" if ctx.world == 1 or args.exchange == "none" else
                                   "one async %s of the [%d, B_local] reward+done slab per chunk, double-buffered "
                                   "(overlaps the next chunk); %d per rank in a timed region"
                                   % (args.exchange, res["chunk"], -(-args.steps // res["chunk"]))},
            "steady_state": res["steady_state"], "sustained": sustained,
            "reps_ms": res["reps_ms"], "timing": "median of %d repetitions of the %d-step region, each bracketed by "
                                                 "barrier + synchronize, max over ranks" % (len(res["reps_ms"]), args.steps),
            "pair_steps_per_sec": res["value"] * N,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": res["kernel"], "bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": launch_ms,
                         "avg_launch_ms_is": "hipEvent span of the timed region on the launch stream / launches: an upper "
                                             "bound on the kernel's duration (includes inter-launch gaps%s)"
                                             % (" and stream waits on the collective" if ctx.world > 1 else ""),
                         "bracketed_launch_ms": res.get("bracketed_launch_ms"), "sampled_launches": res.get("sampled_launches"),
                         "copy_same_bytes_ms": ccopy, "frac_of_copy": (ccopy / launch_ms) if ccopy else None,
                         "torch_copy_same_bytes_ms": ccopy_torch,
                         "hbm_resident": resident},
        }
        if sec is not None:
            out[secondary] = {k: sec[k] for k in ("scaling", "variant", "envs_per_gpu", "envs_total", "chunk", "value",
                                                  "ms_per_step", "reps_ms", "kernel", "steps_per_launch",
                                                  "stream_span_ms_per_step", "steady_state")}
            out[secondary]["workload"] = workloads[secondary]
            sms = sec["stream_span_ms_per_step"] * max(1.0, sec["steps_per_launch"])
            out[secondary]["launch_bytes"] = sec["launch_bytes"]
            out[secondary]["roofline_frac"] = sec["launch_bytes"] / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if fused is not None:
            fms = fused["stream_span_ms_per_step"] * max(1.0, fused["steps_per_launch"])
            out["fused_rollout"] = {k: fused[k] for k in ("value", "ms_per_step", "reps_ms", "kernel", "steps_per_launch",
                                                          "chunk", "launch_bytes")}
            out["fused_rollout"]["roofline_frac"] = fused["launch_bytes"] / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["fused_rollout"]["note"] = ("same workload, each chunk of %d steps fused into ONE launch of the looped kernel "
                                            "(uam_config.fuse_rollout = always): open-loop action tape only; env state stays in "
                                            "registers for the chunk, so a step moves %d B/env instead of %d"
                                            % (fused["chunk"], fused["launch_bytes"] // (fused["chunk"] * res["envs_per_gpu"]),
                                               res["step_bytes"] // res["envs_per_gpu"]))
        if goal is not None:
            out["goal_policy"] = goal
        if planner is not None:
            out["config5_planner"] = planner
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
        sys.stdout.flush()
    if args.dump_exchange:
        exchange_dump(ctx, args, args.dump_exchange)
    if ctx.world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
