#!/usr/bin/env python3
"""Worker of tests/test_gpu_rccl.py (a process of its own: it initialises a process group).

ONE rank, backend "nccl" (= RCCL on ROCm): the exact calls a rank of the 8-GPU run makes through
uamenv.sharded.RolloutExchange — async `dist.gather(..., dst=0)` / `dist.all_gather_into_tensor` of the uint8 slab that
holds the chunk's float32 rewards and uint8 dones, double-buffered, stream-ordered `wait()` — carried by RCCL on the
one GPU of the box.  The slabs are filled by the HIP env (uam_rollout into the slab views); what the collective
delivered must equal them byte for byte.  Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import uamenv
    from uamenv.sharded import RolloutExchange
    B, N, K = 4096, 8, 12
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "modes": {}}
    for mode in ("gather", "all_gather"):
        env = uamenv.BatchedSingleAircraftEnv(B, N, variant="continuous", seed=3)
        env.reset()
        dev = env.device
        ex = RolloutExchange(dist, K, [B], 0, dev, mode=mode, force=True)
        info = torch.empty((K, B), dtype=torch.uint8, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        want = {}
        checked = 0
        for chunk in range(5):
            c = chunk & 1
            ex.begin(c)                      # stream-ordered wait on the collective that last read this slab
            if chunk >= 2:                   # ... which has delivered chunk - 2: compare before the slab is refilled
                r, d = ex.gathered(c)
                wr, wd = want.pop(chunk - 2)
                assert torch.equal(r, wr) and torch.equal(d, wd), (mode, chunk)
                checked += 1
            rew, done = ex.slab(c)
            tape = (torch.rand((K, B), dtype=torch.float32, device=dev, generator=gen) * 2 - 1).contiguous()
            env.rollout_into(tape, rew, done, info)
            want[chunk] = (rew.clone(), done.clone())
            ex.exchange(c)
            assert ex._pending[c] is not None and hasattr(ex._pending[c], "wait"), "no async work handle"
        ex.drain()
        for chunk in (3, 4):
            r, d = ex.gathered(chunk & 1)
            wr, wd = want.pop(chunk)
            assert torch.equal(r, wr) and torch.equal(d, wd), (mode, chunk)
            checked += 1
        torch.cuda.synchronize()
        assert ex.collectives == 5 and ex.mode == mode, (ex.collectives, ex.mode)      # no silent fallback to all_gather
        out["modes"][mode] = {"collectives": ex.collectives, "chunks_checked": checked,
                              "slab_bytes": int(ex._slabs[0][0].numel())}
        env.close()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
