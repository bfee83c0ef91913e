"""The RCCL (`nccl`) code path of the sharded run on the one-GPU box: a one-rank process group carries the chunk's
reward/done slab through the same RolloutExchange calls a rank of the 8-GPU run makes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_nccl_group_carries_the_slab_in_both_modes():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py")], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1
    for mode in ("gather", "all_gather"):
        assert d["modes"][mode]["collectives"] == 5 and d["modes"][mode]["chunks_checked"] == 5
