"""N>1 path on CPU: world_size-2 `gloo` processes exercise the shard partition and the chunked all-gather
of rewards/dones.  The local stepper is the CPU oracle injected by the test (the product has no CPU
backend); what is under test is the host-side sharding logic that bench.py / the 8-GPU run rely on:
global-index Philox streams make the union of the shards bit-identical to one unsharded batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStepper:
    """VecEnv-shaped adapter over the CPU oracle, torch CPU tensors in/out (test-only)."""

    def __init__(self, num_envs, num_intruders, env_id_offset, seed):
        from oracle import oracle
        self.o = oracle
        cfg = oracle.default_config(num_envs, num_intruders, seed=seed, env_id_offset=env_id_offset,
                                    rng_mode=oracle.RNG_PHILOX, precision=oracle.POS_F32, auto_reset=1)
        self.b = oracle.OracleBatch(cfg, oracle.MATH_PORTABLE)
        self.num_envs = num_envs

    def reset(self):
        return torch.from_numpy(self.b.reset().astype(np.float32))

    def step(self, actions):
        obs, rew, done, info = self.b.step(np.asarray(actions, dtype=np.int32))
        return (torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(rew.astype(np.float32)),
                torch.from_numpy(done), info)

    def close(self):
        self.b.close()


def _worker(rank, world, port, G, N, T, chunk, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uamenv.sharded import ShardedSingleAircraftEnv
    env = ShardedSingleAircraftEnv(G, N, dist=dist, chunk=chunk,
                                   env_factory=lambda n, off, dev: OracleStepper(n, N, off, seed=77))
    env.reset()
    rs = np.random.RandomState(5)
    acts = rs.randint(0, 9, size=(T, G)).astype(np.int32)
    gathered_r, gathered_d = [], []
    for t in range(T):
        env.step(acts[t, env.first:env.first + env.num_envs])
        if env.chunk_full() or t == T - 1:
            r, d = env.gather_chunk()
            gathered_r.append(r.numpy())
            gathered_d.append(d.numpy())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rew=np.concatenate(gathered_r), done=np.concatenate(gathered_d),
             first=env.first, count=env.num_envs)
    env.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("G", [64, 37])   # even and uneven partitions
def test_two_rank_shards_equal_one_batch(tmp_path, G):
    N, T, chunk, world = 8, 48, 20, 2
    port = 29600 + (os.getpid() % 300) + G
    mp.spawn(_worker, args=(world, port, G, N, T, chunk, str(tmp_path)), nprocs=world, join=True)
    # single-process reference over the whole batch, same seed and actions
    sys.path.insert(0, ROOT)
    whole = OracleStepper(G, N, 0, seed=77)
    whole.reset()
    rs = np.random.RandomState(5)
    acts = rs.randint(0, 9, size=(T, G)).astype(np.int32)
    rew, done = [], []
    for t in range(T):
        _, r, d, _ = whole.step(acts[t])
        rew.append(r.numpy())
        done.append(d.numpy())
    rew, done = np.stack(rew), np.stack(done)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert sum(int(o["count"]) for o in outs) == G and int(outs[1]["first"]) == int(outs[0]["count"])
    for o in outs:   # every rank holds the full gathered slabs, identical to the unsharded run
        assert o["rew"].shape == (T, G)
        assert np.array_equal(o["rew"], rew)
        assert np.array_equal(o["done"], done)


def test_shard_bounds_partition():
    from uamenv.sharded import shard_bounds
    for G in (1, 7, 64, 65536, 65537):
        for W in (1, 2, 3, 8):
            spans = [shard_bounds(G, W, r) for r in range(W)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == G
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


# ---------------------------------------------------------------------------------------------------------------
# bench.py's own N>1 path: RolloutExchange (double-buffered [K, B_local] reward/done slabs, ONE collective per
# chunk).  On CPU the slabs are filled by the oracle instead of uam_rollout; everything else — slab layout, the
# alternation, begin/exchange/drain ordering, partial chunks, uneven shards, gather-to-rank-0 — is the code the
# 8-GPU run executes.
def _exchange_worker(rank, world, port, G, N, T, K, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uamenv.sharded import RolloutExchange, shard_bounds
    spans = [shard_bounds(G, world, r) for r in range(world)]
    counts = [c for _, c in spans]
    first, count = spans[rank]
    env = OracleStepper(count, N, first, seed=31)
    env.reset()
    rs = np.random.RandomState(9)
    acts = rs.randint(0, 9, size=(T, G)).astype(np.int32)
    ex = RolloutExchange(dist, K, counts, rank, torch.device("cpu"), mode=mode)
    got_r, got_d, chunks = [], [], []
    t, c = 0, 0
    while t < T:
        k = min(K, T - t)
        ex.begin(c)
        if c >= 2 and (mode == "all_gather" or rank == 0):      # slab c was exchanged two chunks ago: read it before reuse
            r, d = ex.gathered(c, chunks[c - 2])
            got_r.append(r.clone().numpy()); got_d.append(d.clone().numpy())
        rew, done = ex.slab(c)
        assert rew.shape == (K, count) and rew.is_contiguous() and done.is_contiguous()
        for j in range(k):
            _, r_, d_, _ = env.step(acts[t + j, first:first + count])
            rew[j].copy_(r_)
            done[j].copy_(d_.to(torch.uint8))
        ex.exchange(c)
        chunks.append(k)
        t += k
        c += 1
    ex.drain()
    if mode == "all_gather" or rank == 0:
        for cc in range(max(0, c - 2), c):
            r, d = ex.gathered(cc, chunks[cc])
            got_r.append(r.clone().numpy()); got_d.append(d.clone().numpy())
        np.savez(os.path.join(out_dir, "x%d.npz" % rank), rew=np.concatenate(got_r), done=np.concatenate(got_d))
    assert ex.collectives == c
    env.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("G,mode", [(64, "all_gather"), (37, "all_gather"), (37, "gather")])
def test_rollout_exchange_two_ranks_equal_one_batch(tmp_path, G, mode):
    N, T, K, world = 8, 50, 16, 2         # 50 = 3 full chunks + a partial one of 2 steps
    port = 29950 + (os.getpid() % 300) + G + (7 if mode == "gather" else 0)
    mp.spawn(_exchange_worker, args=(world, port, G, N, T, K, mode, str(tmp_path)), nprocs=world, join=True)
    whole = OracleStepper(G, N, 0, seed=31)
    whole.reset()
    rs = np.random.RandomState(9)
    acts = rs.randint(0, 9, size=(T, G)).astype(np.int32)
    rew, done = [], []
    for t in range(T):
        _, r, d, _ = whole.step(acts[t])
        rew.append(r.numpy()); done.append(d.numpy())
    rew, done = np.stack(rew), np.stack(done).astype(np.uint8)
    receivers = range(world) if mode == "all_gather" else [0]
    for r in receivers:
        o = np.load(os.path.join(str(tmp_path), "x%d.npz" % r))
        assert o["rew"].shape == (T, G)
        assert np.array_equal(o["rew"], rew) and np.array_equal(o["done"], done)
    if mode == "gather":
        assert not os.path.exists(os.path.join(str(tmp_path), "x1.npz"))


def _force_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uamenv.sharded import RolloutExchange
    K, B = 6, 37
    ok = True
    for mode in ("gather", "all_gather"):
        idle = RolloutExchange(dist, K, [B], 0, torch.device("cpu"), mode=mode)              # a one-rank world: no collective
        idle.exchange(0)
        ok = ok and idle.collectives == 0
        ex = RolloutExchange(dist, K, [B], 0, torch.device("cpu"), mode=mode, force=True)    # ... unless forced
        g = torch.Generator().manual_seed(3)
        for c in range(3):
            ex.begin(c)
            rew, done = ex.slab(c)
            rew.copy_(torch.randn((K, B), generator=g))
            done.copy_(torch.randint(0, 2, (K, B), generator=g).to(torch.uint8))
            want = (rew.clone(), done.clone())
            ex.exchange(c)
            ex.begin(c)
            r, d = ex.gathered(c)
            ok = ok and torch.equal(r, want[0]) and torch.equal(d, want[1])
        ok = ok and ex.collectives == 3 and ex.mode == mode
    open(os.path.join(out_dir, "force_ok"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_forced_exchange_in_a_one_rank_world(tmp_path):
    """RolloutExchange(force=True): the collective is issued even in a world of one rank — what the `-m gpu` test
    tests/test_gpu_rccl.py relies on to drive the RCCL calls on a one-GPU box."""
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_force_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert open(os.path.join(str(tmp_path), "force_ok")).read() == "1"


def test_bench_self_launch_spawns_ranks_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE must start torch.distributed.run as a CHILD and must not have
    imported torch (let alone touched the GPU) in the parent."""
    import importlib
    sys.path.insert(0, ROOT)
    import bench
    importlib.reload(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        seen["env"] = env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7                                     # the child's status is the parent's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def parse_args")]
    assert "import torch" not in head and "import uamenv" not in head   # nothing GPU-capable at module import


def test_bench_cpu_baseline_with_a_fractional_cgroup_quota(monkeypatch):
    """bench.py's cpu_baseline on a box whose cgroup grants a fractional number of CPUs (cpu.max "1650000 100000"): the
    thread counts it tries must be integers (a float reached the oracle's ctypes binding once and the driver's line
    carried `cpu_baseline.value: null`), `value` is the best of them and `cores` names it."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "cpu_quota", lambda: 2.6)
    out = bench.cpu_baseline(256, 4, 2, scale=0.02)
    assert out["value"] and out["value"] > 0 and out["kind"] == "port" and out["single_thread"] > 0
    assert isinstance(out["cores"], int) and str(out["cores"]) in out["by_threads"]
    assert out["value"] == max(out["by_threads"].values()) and out["cgroup_cpu_quota"] == 2.6
