#!/bin/bash
# Development tool: HBM traffic (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes) of the step
# kernel for other variants than the headline one; prints one JSON object.  Run on the GPU box via gpurun.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_variants
mkdir -p $OUT
cd /tmp
for v in her9 randintruder random_start; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${v}_$c -- python3 $R/bench.py --no-build --sustain-seconds 0 --no-secondary --steps 60 --warmup 20 --no-cpu-baseline --variant $v > $OUT/${v}_$c.log 2>&1
  done
done
cd $R
python - <<PY
import csv, glob, json
out = {}
for v in ("her9", "randintruder", "random_start"):
    acc = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for c in acc:
        for f in glob.glob("$OUT/%s_%s/*/*counter_collection.csv" % (v, c)):
            for r in csv.DictReader(open(f)):
                if "uam_sweep_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[c].append(float(r["Counter_Value"]))
    fetch = sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"]))
    write = sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"]))
    bench = [json.loads(l) for l in open("$OUT/%s_FETCH_SIZE.log" % v) if l.startswith(chr(123) + chr(34) + "metric")][-1]
    out[v] = {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
              "launches": len(acc["FETCH_SIZE"]), "note": "65536 envs x 32 intruders; FETCH_SIZE doubled (gfx950), KiB units"}
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
