#!/bin/bash
# Development tool: SQ counters of the looped kernel on one shape (run on the GPU box via gpurun).
# usage: tools/pmc_many.sh OUTDIR B N K [variant]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/many_probe.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "uam_sweep" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        acc[k]["_vgpr"].append(float(r["VGPR_Count"])); acc[k]["_lds"].append(float(r["LDS_Block_Size"])); acc[k]["_scratch"].append(float(r["Scratch_Size"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
