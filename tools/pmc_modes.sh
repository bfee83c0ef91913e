#!/bin/bash
# Development tool: HBM traffic (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes) of the step and
# K-step kernels of the exact-position modes at 65536 x 32; prints one JSON object.  Run on the GPU box via gpurun.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_modes
mkdir -p $OUT
cd /tmp
for m in "mt19937" "philox"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${m}_$c -- python3 $R/bench.py --no-build --steps 64 --warmup 32 --reps 2 --no-cpu-baseline --sustain-seconds 0 --rng $m --precision exact > $OUT/${m}_$c.log 2>&1
  done
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out = {}
for m in ("mt19937", "philox"):
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/%s_%s/*/*counter_collection.csv" % (m, c)):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "uam_sweep" not in k or int(r["Grid_Size"]) != 65536 * 8: continue
                pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in pm.items():
        if not cs["FETCH_SIZE"] or not cs["WRITE_SIZE"]: continue
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        out["%s exact: %s" % (m, k)] = {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "launches": len(cs["FETCH_SIZE"])}
out["_note"] = "65536 envs x 32 intruders, exact positions; algorithmic bytes per step 127 270 912 (x 32 for a K-step launch of 32 steps); FETCH_SIZE doubled (gfx950), KiB units, separate --pmc passes; Infinity-Cache hits are counted"
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
