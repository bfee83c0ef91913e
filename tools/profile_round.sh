#!/bin/bash
# Produces the per-round evidence under gpurun_out/<tag>/ on the GPU box:
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python bench.py --steps 500`
#   pmc_summary.json        FETCH_SIZE / WRITE_SIZE (separate passes) per kernel, gfx950-corrected
# Copy what should be judged into profiles/ afterwards.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
cd $R
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uam_" not in k: continue
        name = "uam_sweep_kernel" if "sweep" in k else "uam_own_kernel" if "own" in k else "uam_noise_kernel" if "noise" in k else "uam_reset_kernel"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    fetch = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"]))     # KB per launch as reported
    write = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"]))
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half of a wide coalesced read on gfx950 -> x2; units are KiB
    out[k] = {"FETCH_SIZE_KB_reported": fetch, "WRITE_SIZE_KB": write, "launches": len(cs["FETCH_SIZE"]),
              "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
              "note": "FETCH_SIZE doubled per the gfx950 correction; separate --pmc passes; 16-B-per-lane accesses"}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
head -6 $OUT/kernel_stats.csv | cut -c1-160
