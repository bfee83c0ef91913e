#!/bin/bash
# Development tool: SQ instruction-mix / wait counters of the MCTS search kernel (run via gpurun).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc_mcts}
mkdir -p $OUT
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python $R/tools/bench_mcts.py --reps 1 --cpu-roots 0 ${@:2} > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if "mcts_search" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
