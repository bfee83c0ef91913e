#!/usr/bin/env python3
"""Development tool: the library's copy kernel (uam_bench_copy) in its shapes against torch's copy, at the sizes bench.py
uses it (94 MB and 1.5 GB moved per copy: half read, half written)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uamenv import _capi as capi
L = capi.lib()
dev = torch.device("cuda", 0)
for total in (93716480, 1499463680):
    n = (total // 2) // 16 * 16
    src = torch.zeros((n,), dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for shape in ("torch", "0", "1", "512", "1024", "4096", "8192", "16384"):
        if shape != "torch":
            os.environ["UAMENV_COPY_SHAPE"] = shape
        f = (lambda: dst.copy_(src)) if shape == "torch" else (lambda: capi.check(L.uam_bench_copy(0, src.data_ptr(), dst.data_ptr(), n, st)))
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print("moved %.0f MB shape %-6s %.4f ms  %.2f TB/s" % (2 * n / 1e6, shape, ms, 2 * n / ms / 1e9))
