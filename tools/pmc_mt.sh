#!/bin/bash
# Development tool: SQ issue / wait counters of the MT19937 step kernels (run via gpurun).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc_mt}
mkdir -p $OUT
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  UAMENV_MANY=off rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/tools/mt_probe.py > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
cd $R
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "sweep_mt" in k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "sweep_mt" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        d = sum(dur[k]) / max(1, len(dur[k]))
        out = {c: round(sum(v)/len(v), 1) for c, v in cs.items()}
        if "SQ_ACTIVE_INST_VALU" in out:
            out["valu_busy_per_simd"] = round(out["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * d * 1e-6 * 2.4e9), 3)
        print(k, "avg_us %.1f" % d, out)
PY
