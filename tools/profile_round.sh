#!/bin/bash
# Produces the per-round evidence under gpurun_out/<tag>/ on the GPU box (copy what should be judged into profiles/):
#   kernel_stats.csv          rocprofv3 --kernel-trace --stats of `python3 bench.py --no-secondary --no-cpu-baseline`
#                             (the primary workload alone, so that the step kernel's average is one shape's average)
#   kernel_by_grid.csv        the DEFAULT command (`python3 bench.py --no-cpu-baseline`: primary, strong, fused and the
#                             1 M-env pass) traced, durations grouped by kernel and grid size
#   pmc_summary.json          FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of the default command per
#                             kernel and grid size, gfx950-corrected (FETCH_SIZE x2), bytes per launch
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-build --sustain-seconds 0 --no-secondary --no-cpu-baseline > $OUT/bench_primary_under_rocprof.json 2> $OUT/trace.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
echo "trace 1 done"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_all -- python3 $R/bench.py --no-build --sustain-seconds 0 --no-cpu-baseline > $OUT/bench_default_under_rocprof.json 2> $OUT/trace_all.log
echo "trace 2 done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --no-build --sustain-seconds 0 --steps 60 --warmup 20 --reps 2 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  echo "pmc $c done"
done
cd $R
python3 - <<PY
import csv, glob, json, collections
def short(k):
    return k.split("(")[0].replace("void ", "")
# durations by kernel and grid of the default command
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/trace_all/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "uam_" not in r["Kernel_Name"]: continue
        acc[(short(r["Kernel_Name"]), int(r.get("Grid_Size") or r["Grid_Size_X"]), int(r.get("Workgroup_Size") or r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/kernel_by_grid.csv", "w") as f:
    f.write("kernel,grid_threads,block,calls,avg_us,min_us,max_us\n")
    for (k, g, b), v in sorted(acc.items()):
        f.write('"%s",%d,%d,%d,%.3f,%.3f,%.3f\n' % (k, g, b, len(v), sum(v) / len(v), min(v), max(v)))
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "uam_" not in r["Kernel_Name"]: continue
        pm[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (k, g), cs in sorted(pm.items()):
    if not cs["FETCH_SIZE"] or not cs["WRITE_SIZE"]: continue
    fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])     # KiB per launch as reported
    write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    out["%s grid=%d" % (k, g)] = {"FETCH_SIZE_KB_reported": fetch, "WRITE_SIZE_KB": write, "launches": len(cs["FETCH_SIZE"]),
                                  "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
out["_note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); separate --pmc passes with "
                "--kernel-trace only; Infinity-Cache hits are counted")
# the entry bench.py reads: the step kernel at the headline grid (65536 envs x 8 lanes)
for key, v in list(out.items()):
    if key.startswith("uam_sweep_kernel<false, 0, 0, false, false> grid=%d" % (65536 * 8)) and "uam_sweep_kernel" not in out:
        out["uam_sweep_kernel"] = v
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
head -8 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/kernel_by_grid.csv
