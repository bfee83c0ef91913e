#!/bin/bash
# Development tool: SPI resource-stall counters + wave stats of the step kernel (GPU box).  usage: pmc_spi.sh OUT B N K variant
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SPI_RA_[A-Z_0-9]*\|SQ_WAVES_[A-Z_0-9]*\|SQ_LEVEL_WAVES\|SQ_BUSY_CU_CYCLES\|SPI_CSN_[A-Z_0-9]*" | sort -u > $OUT/avail.txt
i=0
for set in "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN" "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_BAR_CU_FULL_CSN" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/many_probe.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
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
        acc[k]["_vgpr"].append(float(r["VGPR_Count"])); acc[k]["_agpr"].append(float(r["Accum_VGPR_Count"])); acc[k]["_lds"].append(float(r["LDS_Block_Size"])); acc[k]["_scratch"].append(float(r["Scratch_Size"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
