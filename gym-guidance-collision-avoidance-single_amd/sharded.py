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
