"""bench.py's one-line JSON contract (driver-facing), on a tiny run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "20",
                          "--batch", "4096", "--intruders", "8", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert abs(d["value"] - 4096 * 60 / (d["ms_per_step"] * 1e-3 * 60)) / d["value"] < 1e-6
    # median of the listed repetitions; the second workload and the 1 M-env roofline pass ride in the same line
    reps = sorted(d["reps_ms"])
    assert len(reps) == 5 and abs(d["ms_per_step"] * 60 - reps[2]) < 1e-9
    assert d["strong"]["variant"] == "continuous" and d["strong"]["envs_total"] == 4096 and d["strong"]["value"] > 0
    hr = r["hbm_resident"]
    assert hr["envs"] == 1 << 20 and 0 < hr["frac"] < 1 and hr["copy_same_bytes_ms"] > 0
    # both action regimes of SURVEY 8d: the random tape is `value`, the goal-seeking closed loop rides beside it
    gp = d["goal_policy"]
    assert gp["value"] > 0 and gp["episode_ends_per_env_step"] > 0 and 0 < gp["roofline_frac"] < 1, gp
