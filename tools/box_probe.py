#!/usr/bin/env python3
"""Development probe for the box-to-box spread of the HBM-resident figure: on ONE lease, the 1 M-env step (1.5 GB per launch), the
library's flat copy of the same bytes, and what rocm-smi says about clocks / power / temperature WHILE the step loop runs.
Prints one JSON line.  usage: box_probe.py"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uamenv
from uamenv import _capi as capi

B = 1 << 20
# PROBE_PRE: what the process does to the device memory before the env is created (does the placement of the env's arrays
# matter?): "churn" = allocate and free 200 GB first; "hold" = keep 150 GB allocated while the env lives
pre = os.environ.get("PROBE_PRE", "")
held = None
if pre == "churn":
    x = torch.empty((200 * (1 << 30),), dtype=torch.uint8, device="cuda:0"); x.zero_(); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
elif pre == "hold":
    held = torch.empty((150 * (1 << 30),), dtype=torch.uint8, device="cuda:0"); held[::1 << 21].zero_(); torch.cuda.synchronize()
env = uamenv.BatchedSingleAircraftEnv(B, 32, seed=0, fuse_rollout="never")
env.reset()
dev = env.device
tape = torch.randint(0, 9, (8, B), dtype=torch.int32, device=dev)
rew = torch.empty((8, B), dtype=torch.float32, device=dev); done = torch.empty((8, B), dtype=torch.uint8, device=dev)
info = torch.empty((8, B), dtype=torch.uint8, device=dev)
for _ in range(12):
    env.rollout_into(tape, rew, done, info)
torch.cuda.synchronize()
smi = {}


def sample():
    time.sleep(1.0)
    try:
        smi["out"] = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
    except Exception as ex:
        smi["out"] = repr(ex)


th = threading.Thread(target=sample); th.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); times = []
while time.time() - t0 < 4.0:
    e0.record()
    for _ in range(8):
        env.rollout_into(tape, rew, done, info)
    e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / 64)
th.join()
times.sort()
step_ms = times[len(times) // 2]
nbytes = env.step_bytes
n = (nbytes // 2) // 16 * 16
src = torch.zeros((n,), dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    capi.check(L.uam_bench_copy(0, src.data_ptr(), dst.data_ptr(), n, st))
e0.record()
for _ in range(30):
    capi.check(L.uam_bench_copy(0, src.data_ptr(), dst.data_ptr(), n, st))
e1.record(); torch.cuda.synchronize()
copy_ms = e0.elapsed_time(e1) / 30
try:
    ident = subprocess.run(["rocm-smi", "--showuniqueid", "--showbus"], capture_output=True, text=True, timeout=20).stdout
    ident = [l.split(":", 1)[-1].strip() for l in ident.splitlines() if "Unique ID" in l or "PCI Bus" in l]
except Exception as ex:
    ident = [repr(ex)]
keep = [l.strip() for l in smi.get("out", "").splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power", "Temperature", "Performance Level"))]
print(json.dumps({"host": os.uname().nodename, "gpu": ident, "pre": pre, "step_ms_1M_envs": step_ms, "step_frac_of_8TBps": nbytes / (step_ms * 1e-3) / 8e12,
                  "flat_copy_ms_same_bytes": copy_ms, "copy_frac_of_8TBps": 2 * n / (copy_ms * 1e-3) / 8e12,
                  "step_frac_of_copy": copy_ms / step_ms, "rocm_smi_under_load": keep}))
