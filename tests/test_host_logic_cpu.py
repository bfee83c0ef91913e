"""Host-side logic that needs no GPU: bench.py's profiler detection and the example collector's return arithmetic."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_bench_detects_a_profiler_preload(monkeypatch):
    """A process under rocprofv3 (--pmc initialises the GPU before Python starts) must not spawn `make` or run the
    seconds-long closing pass: bench.py decides that from the environment the profiler leaves."""
    bench = importlib.import_module("bench")
    for k in list(os.environ):
        if k.startswith("ROCPROF"):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert bench.under_profiler()


def test_example_gae_matches_the_reference_runner_loop():
    """examples/ppo_rollout.gae_returns against the reverse loop of baselines/ppo2/runner.py:46-58 restated in NumPy
    (float32, done flags reported BEFORE each step, bootstrap from the last value), and sf01's env-major flattening."""
    import torch
    import ppo_rollout as pr
    rs = np.random.RandomState(0)
    T, B, gamma, lam = 17, 9, 0.99, 0.95
    rew = rs.normal(size=(T, B)).astype(np.float32)
    val = rs.normal(size=(T, B)).astype(np.float32)
    dn = rs.uniform(size=(T, B)) < 0.2
    last_d = rs.uniform(size=B) < 0.2
    last_v = rs.normal(size=B).astype(np.float32)
    adv = np.zeros_like(rew)
    lastgaelam = np.zeros(B, dtype=np.float32)
    for t in reversed(range(T)):
        nnt = 1.0 - (last_d if t == T - 1 else dn[t + 1]).astype(np.float32)
        nv = last_v if t == T - 1 else val[t + 1]
        delta = rew[t] + np.float32(gamma) * nv * nnt - val[t]
        adv[t] = lastgaelam = delta + np.float32(gamma * lam) * nnt * lastgaelam
    ret, a = pr.gae_returns(torch.from_numpy(rew), torch.from_numpy(val), torch.from_numpy(dn), torch.from_numpy(last_d),
                            torch.from_numpy(last_v), gamma, lam)
    np.testing.assert_allclose(a.numpy(), adv, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ret.numpy(), adv + val, rtol=1e-5, atol=1e-6)
    x = torch.arange(T * B * 2).view(T, B, 2)
    assert torch.equal(pr.sf01(x), torch.from_numpy(np.ascontiguousarray(x.numpy().swapaxes(0, 1)).reshape(T * B, 2)))
