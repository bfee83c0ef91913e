"""Env batch sharded across the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm).

The path shards naturally — envs are independent — so a rank simply owns the contiguous
slice [rank*B_local, (rank+1)*B_local) of the global batch.  `env_id_offset` makes the Philox streams a
function of the GLOBAL env index, so the union of the shards is bit-identical to one big batch.  No state
is ever exchanged; the only collective is the gather of the reward/done slabs, issued once per rollout
chunk (latency-bound: 5 B/env/step) rather than per step.
"""
import numpy as np

from .envs import BatchedSingleAircraftEnv


def shard_bounds(global_envs, world_size, rank):
    """Contiguous, balanced partition of `global_envs` envs: (first global index, count) for `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(global_envs), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


class RolloutExchange:
    """The one exchange of the sharded path: per rollout chunk of K steps, the rank's `[K, B_slab]`
    reward (float32) and done (uint8) slabs leave in ONE collective.  Both live in one byte buffer, so a single
    all-gather (or gather to rank 0) moves them; two slabs alternate, so the collective of chunk c runs on the
    backend's own stream / thread while chunk c+1 steps into the other slab.

    Every rank contributes the same number of bytes, sized for the widest shard (`max(counts)`); a narrower
    rank's `[K, count]` slabs are contiguous at the front of their halves and the tail is padding.  Works on device tensors over RCCL
    (backend "nccl") and on CPU tensors over gloo (the world_size-2 tests).  This is the object `bench.py` steps
    into, and the parallel step of baselines/common/vec_env/subproc_vec_env.py:63-74 (`step_wait`: one result per
    worker, stacked) turned inside out: state never moves, only 5 bytes per env-step do.

    `force=True` issues the collective even in a world of ONE rank (a one-rank process group): the call sequence a rank
    of the 8-GPU run makes — async gather / all_gather_into_tensor of the uint8 slab, stream-ordered wait — then runs
    through RCCL on a one-GPU box (tests/test_gpu_rccl.py)."""

    def __init__(self, dist, K, counts, rank, device, mode="all_gather", force=False):
        import torch
        if mode not in ("all_gather", "gather", "none"):
            raise ValueError("mode must be all_gather, gather or none")
        self.dist, self.K, self.counts, self.rank, self.mode = dist, int(K), list(counts), int(rank), mode
        self.world = len(self.counts)
        self.B_slab = max(self.counts)
        self.count = self.counts[self.rank]
        KB = (self.K * self.B_slab + 15) & ~15     # float32 views of every rank's segment stay 4-byte aligned
        self._KB = KB

        def make():
            raw = torch.zeros((KB * 5,), dtype=torch.uint8, device=device)
            n = self.K * self.count
            return (raw, raw[:n * 4].view(torch.float32).view(self.K, self.count),
                    raw[KB * 4:KB * 4 + n].view(self.K, self.count))

        self._slabs = [make(), make()]
        self._pending = [None, None]
        self._active = (self.world > 1 or bool(force)) and mode != "none"
        receives = self._active and (mode == "all_gather" or (mode == "gather" and self.rank == 0))
        self._gathered = [torch.zeros((self.world * KB * 5,), dtype=torch.uint8, device=device) for _ in range(2)] \
            if receives else None
        self.collectives = 0

    def slab(self, c):
        """(reward float32[K, count], done uint8[K, count]) contiguous views of slab c (0/1)."""
        _, rew, done = self._slabs[c & 1]
        return rew, done

    def begin(self, c):
        """Before stepping into slab c: its previous collective must have read it (stream-ordered on the device —
        the host does not block with RCCL)."""
        w = self._pending[c & 1]
        if w is not None:
            w.wait()
            self._pending[c & 1] = None

    def exchange(self, c):
        """Start the collective of slab c (asynchronous)."""
        if not self._active:
            return
        raw = self._slabs[c & 1][0]
        if self.mode == "all_gather":
            self._pending[c & 1] = self.dist.all_gather_into_tensor(self._gathered[c & 1], raw, async_op=True)
        else:
            outs = list(self._gathered[c & 1].view(self.world, -1).unbind(0)) if self.rank == 0 else None
            try:
                self._pending[c & 1] = self.dist.gather(raw, gather_list=outs, dst=0, async_op=True)
            except (RuntimeError, NotImplementedError) as ex:
                # a backend without gather (raised at call time, on every rank alike): the all-gather moves the same
                # slabs — every rank receives them instead of rank 0 only
                import torch
                import warnings
                warnings.warn("RolloutExchange: gather unavailable (%s); falling back to all_gather" % (ex,))
                self.mode = "all_gather"
                if self._gathered is None:
                    self._gathered = [torch.zeros((self.world * self._KB * 5,), dtype=torch.uint8, device=raw.device)
                                      for _ in range(2)]
                self._pending[c & 1] = self.dist.all_gather_into_tensor(self._gathered[c & 1], raw, async_op=True)
        self.collectives += 1

    def drain(self):
        for c in (0, 1):
            self.begin(c)

    def gathered(self, c, k=None):
        """After `begin(c)` / `drain()`: (rewards float32[k, global_envs], dones uint8[k, global_envs]) of the chunk
        last exchanged from slab c, ranks concatenated in global env order (padding columns dropped)."""
        import torch
        k = self.K if k is None else int(k)
        if not self._active:
            rew, done = self.slab(c)
            return rew[:k], done[:k]
        if self._gathered is None:
            raise RuntimeError("this rank receives nothing in mode %r" % self.mode)
        g = self._gathered[c & 1].view(self.world, self._KB * 5)
        rews, dones = [], []
        for r, n in enumerate(self.counts):
            rews.append(g[r, :self.K * n * 4].view(torch.float32).view(self.K, n)[:k])
            dones.append(g[r, self._KB * 4:self._KB * 4 + self.K * n].view(self.K, n)[:k])
        return torch.cat(rews, dim=1), torch.cat(dones, dim=1)


class ShardedSingleAircraftEnv:
    """VecEnv-shaped front end of this rank's shard + chunked all-gather of rewards/dones.

    `env_factory(num_envs, env_id_offset, device)` builds the local stepper; the default is the HIP env.
    (Tests inject a CPU stepper to exercise the partition/gather logic under the gloo backend.)
    """

    def __init__(self, global_envs, num_intruders=None, dist=None, device=None, chunk=32, env_factory=None, **env_kw):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.global_envs = int(global_envs)
        self.first, self.num_envs = shard_bounds(global_envs, self.world, self.rank)
        if self.num_envs < 1:
            raise ValueError("rank %d would own no env (%d envs over %d ranks)" % (self.rank, global_envs, self.world))
        counts = [shard_bounds(global_envs, self.world, r)[1] for r in range(self.world)]
        self._even = len(set(counts)) == 1
        self._counts = counts
        if env_factory is None:
            def env_factory(n, off, dev):
                return BatchedSingleAircraftEnv(n, num_intruders, device=dev, env_id_offset=off, **env_kw)
        self.env = env_factory(self.num_envs, self.first, device)
        self.chunk = int(chunk)
        self._k = 0
        self._rew = None
        self._done = None

    def reset(self):
        self._k = 0
        return self.env.reset()

    def step(self, actions):
        """Step the local shard; rewards/dones are also appended to the current chunk slab."""
        import torch
        obs, rew, done, infos = self.env.step(actions)
        if self._rew is None:
            self._rew = torch.empty((self.chunk, self.num_envs), dtype=torch.float32, device=rew.device)
            self._done = torch.empty((self.chunk, self.num_envs), dtype=torch.uint8, device=rew.device)
        self._rew[self._k].copy_(rew)
        self._done[self._k].copy_(done.to(torch.uint8) if done.dtype != torch.uint8 else done)
        self._k += 1
        return obs, rew, done, infos

    def chunk_full(self):
        return self._k >= self.chunk

    def gather_chunk(self):
        """All-gather the [k, B_local] reward/done slabs of the current chunk.  Returns
        (rewards [k, global_envs] float32, dones [k, global_envs] uint8) on every rank and starts a new chunk."""
        import torch
        k = self._k
        rew, done = self._rew[:k].contiguous(), self._done[:k].contiguous()
        self._k = 0
        if self.world == 1:
            return rew, done
        if self._even:
            rew_all = torch.empty((self.world * k, self.num_envs), dtype=rew.dtype, device=rew.device)
            done_all = torch.empty((self.world * k, self.num_envs), dtype=done.dtype, device=done.device)
            self.dist.all_gather_into_tensor(rew_all, rew)      # rank-major concatenation along dim 0
            self.dist.all_gather_into_tensor(done_all, done)
            return (rew_all.view(self.world, k, self.num_envs).permute(1, 0, 2).reshape(k, self.global_envs),
                    done_all.view(self.world, k, self.num_envs).permute(1, 0, 2).reshape(k, self.global_envs))
        # uneven partition: pad every slab to the widest shard, gather, drop the padding
        cmax = max(self._counts)
        rpad = torch.zeros((k, cmax), dtype=rew.dtype, device=rew.device)
        dpad = torch.zeros((k, cmax), dtype=done.dtype, device=done.device)
        rpad[:, :self.num_envs] = rew
        dpad[:, :self.num_envs] = done
        rew_all = torch.empty((self.world * k, cmax), dtype=rew.dtype, device=rew.device)
        done_all = torch.empty((self.world * k, cmax), dtype=done.dtype, device=done.device)
        self.dist.all_gather_into_tensor(rew_all, rpad)
        self.dist.all_gather_into_tensor(done_all, dpad)
        rew_all, done_all = rew_all.view(self.world, k, cmax), done_all.view(self.world, k, cmax)
        return (torch.cat([rew_all[r, :, :c] for r, c in enumerate(self._counts)], dim=1),
                torch.cat([done_all[r, :, :c] for r, c in enumerate(self._counts)], dim=1))

    def close(self):
        self.env.close()
