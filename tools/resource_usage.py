#!/usr/bin/env python3
"""Summarise csrc/build/resource_usage.txt (`make -C csrc asm`): VGPR / SGPR / scratch / occupancy per kernel."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gym-guidance-collision-avoidance-single_amd/csrc/build/resource_usage.txt"
pat = sys.argv[2] if len(sys.argv) > 2 else "sweep"
cur, rows = None, []
for line in open(path):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, rx in (("sgpr", r"\bSGPRs: (\d+)"), ("vgpr", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                    ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(rx, line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print("%-100s %5s %5s %7s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "occ"))
for r, d in zip(rows, dem):
    if re.search(pat, d):
        print("%-100s %5s %5s %7s %4s" % (d[:100], r.get("vgpr"), r.get("sgpr"), r.get("scratch"), r.get("occ")))
