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
