#!/usr/bin/env python3
"""Development tool: per-wave timeline of ONE step-kernel launch from a `-DUAM_STAMP=1` build.
    UAMENV_STAMP_DUMP=/path/waves.bin UAMENV_LIB=build_ab/stamp.so python bench.py --steps 64 --warmup 32 --reps 1 ...
    python tools/analyze_waves.py /path/waves.bin
Each wave's record: s_memtime cycles per phase of sweep_step (slots 0-8: head, sweep, commit, respawn, reward, heading
changes, per-env outputs, tail/reset, state commit + obs pack) and its start / end in s_memrealtime ticks (10 ns, the same
clock on every CU).  Prints when waves start and end, by position in the grid, by XCD and by phase."""
import sys
import numpy as np

h = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.float64)
nw = h.shape[0]
wpb = int(sys.argv[2]) if len(sys.argv) > 2 else 4            # waves per block
t0 = h[:, 14].min()
st, en = (h[:, 14] - t0) * 0.01, (h[:, 15] - t0) * 0.01
pc = lambda a, q: float(np.percentile(a, q))
print("waves %d   starts: p50 %.2f p99 %.2f max %.2f us   ends: p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f us" %
      (nw, pc(st, 50), pc(st, 99), st.max(), pc(en, 1), pc(en, 10), pc(en, 50), pc(en, 90), pc(en, 99), en.max()))
ph = h[:, :9]
names = ["head", "sweep", "commit", "respawn", "reward", "headings", "outputs", "tail/reset", "store+obs"]
for i, n in enumerate(names):
    print("  %-11s mean %7.0f cycles   corr(end) %+.2f" % (n, ph[:, i].mean(), np.corrcoef(ph[:, i], en)[0, 1] if ph[:, i].std() > 0 else 0.0))
blk = np.arange(nw) // wpb
nb = int(blk.max()) + 1
for q in range(8):
    sel = (blk >= q * nb // 8) & (blk < (q + 1) * nb // 8)
    print("  blocks %5d-%5d: mean start %.2f  mean end %.2f us" % (q * nb // 8, (q + 1) * nb // 8 - 1, st[sel].mean(), en[sel].mean()))
for x in range(8):
    sel = blk % 8 == x
    print("  XCD %d: mean end %.2f  p90 %.2f us" % (x, en[sel].mean(), pc(en[sel], 90)))
slow, fast = en > pc(en, 95), en < pc(en, 20)
print("  slowest 5 %% of waves, cycles per phase:", ph[slow].mean(0).round().astype(int).tolist())
print("  fastest 20 %% of waves, cycles per phase:", ph[fast].mean(0).round().astype(int).tolist())
