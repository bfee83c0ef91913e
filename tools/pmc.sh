#!/bin/bash
# Development tool: per-kernel PMC counters for the bench workload (run on the GPU box via gpurun).
# Separate --pmc passes as MI355X_MICROARCH.md prescribes (TCC slots), kernel-trace only.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/bench.py --no-build --sustain-seconds 0 --no-secondary --steps 40 --warmup 20 --no-cpu-baseline > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if "uam_" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
