# usage: tools/round_evidence.sh TAG   (e.g. r04; results under gpurun_out/TAGev/)
# Per-round evidence beyond tools/profile_round.sh: (1) rocprofv3 durations of the looped kernels at BASELINE config 4's per-GPU
# shard shapes; (2) HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, separate passes) of the RandIntruder step kernel at 65536 x 32
# and 8192 x 64; (3) issue counters of the MCTS search kernel at 8192 x 64 and 2048 x 20.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r04}
O=$R/gpurun_out/${TAG}ev; mkdir -p $O
cd /tmp
for s in "8192 32 256 continuous" "16384 32 128 continuous" "32768 32 64 continuous" "4096 8 256 discrete9"; do
  tag=$(echo $s | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $O/shard_$tag -- python3 $R/tools/many_probe.py $s 30 > $O/shard_$tag.log 2>&1
done
echo "shards done"
for shape in "65536 32" "8192 64"; do
  tag=$(echo $shape | tr ' ' 'x')
  for c in FETCH_SIZE WRITE_SIZE; do
    UAMENV_MANY=off rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/ri_${tag}_$c -- python3 $R/tools/many_probe.py $shape 16 randintruder 12 > $O/ri_${tag}_$c.log 2>&1
  done
done
echo "ri pmc done"
for shape in "--batch 8192 --intruders 64" "--batch 2048 --intruders 20"; do
  tag=$(echo $shape | tr -d ' -' )
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    st=$(echo $set | cut -c1-12 | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/mcts_${tag}_$st -- python3 $R/tools/bench_mcts.py --reps 1 --cpu-roots 0 $shape > $O/mcts_${tag}_$st.log 2>&1 || echo "mcts pass failed"
  done
  python3 $R/tools/bench_mcts.py --reps 5 --cpu-roots 0 $shape 2>/dev/null | grep '^{' >> $O/mcts_bench.jsonl
done
echo "mcts done"
cd $R
python3 - <<PY
import csv, glob, json, collections, os
O = "$O"
out = {}
# (1) shard kernels by grid
acc = collections.defaultdict(list)
for f in glob.glob(O + "/shard_*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "uam_sweep_many" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(k, int(r.get("Grid_Size") or r["Grid_Size_X"]), int(r.get("Workgroup_Size") or r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(O + "/shard_kernel_by_grid.csv", "w") as f:
    f.write("kernel,grid_threads,block,calls,avg_us,min_us,max_us\n")
    for (k, g, b), v in sorted(acc.items()):
        f.write('"%s",%d,%d,%d,%.3f,%.3f,%.3f\n' % (k, g, b, len(v), sum(v) / len(v), min(v), max(v)))
print(open(O + "/shard_kernel_by_grid.csv").read())
# (2) randintruder traffic
ri = {}
for tag, B, N in (("65536x32", 65536, 32), ("8192x64", 8192, 64)):
    cs = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for c in cs:
        for f in glob.glob(O + "/ri_%s_%s/*/*counter_collection.csv" % (tag, c)):
            for r in csv.DictReader(open(f)):
                if "uam_sweep_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    cs[c].append(float(r["Counter_Value"]))
    if cs["FETCH_SIZE"] and cs["WRITE_SIZE"]:
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        hbm = int((2 * fetch + write) * 1024)
        alg = B * (64 * N + 150)
        ri["randintruder %s" % tag] = {"hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch (64 N + 150 per env)": alg, "ratio": round(hbm / alg, 3),
                                       "launches": len(cs["FETCH_SIZE"]), "note": "uam_sweep_kernel<false,0,1,true,false>, one launch per step; FETCH_SIZE doubled (gfx950), KiB units, separate --pmc passes"}
json.dump(ri, open(O + "/pmc_variants.json", "w"), indent=1)
print(json.dumps(ri, indent=1))
# (3) mcts counters
mc = {}
for d in sorted(glob.glob(O + "/mcts_*_SQ*")):
    tag = os.path.basename(d).split("_SQ")[0]
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mcts_search" not in r["Kernel_Name"]: continue
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
            a["_dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            a["_vgpr"].append(float(r["VGPR_Count"]))
        for c, v in a.items():
            mc.setdefault(tag, {})[c] = sum(v) / len(v)
for tag, v in mc.items():
    if "SQ_WAVE_CYCLES" in v:
        v["valu_issue_fraction (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)"] = v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_WAVE_CYCLES"]
        v["any_issue_fraction (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
json.dump(mc, open(O + "/mcts_pmc.json", "w"), indent=1)
print(json.dumps(mc, indent=1)[:2500])
PY
cat $O/mcts_bench.jsonl
# (4) bench lines of the other shapes and modes (no profiler): strong-scaling shards, MT19937 / exact, closed loops
cd $R
for B in 8192 16384 32768; do
  python3 bench.py --no-build --scaling strong --batch $B --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' >> $O/strong_shard_shapes.jsonl
done
python3 bench.py --no-build --scaling strong --batch 4096 --intruders 8 --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' >> $O/strong_shard_shapes.jsonl
python3 bench.py --no-build --rng mt19937 --precision exact --steps 2000 --warmup 200 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' > $O/bench_mt19937_exact.jsonl
python3 bench.py --no-build --precision exact --steps 2000 --warmup 200 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' > $O/bench_philox_exact.jsonl
python3 tools/bench_closed_loop.py 2>/dev/null | grep '^{' > $O/closed_loop.jsonl
python3 - <<PY
import json
for l in open("$O/strong_shard_shapes.jsonl"):
    d = json.loads(l); print("shard", d["config"]["envs_per_gpu"], d["config"]["intruders"], "us/step %.3f" % (d["ms_per_step"] * 1e3), d["config"]["launch"])
for f in ("bench_mt19937_exact", "bench_philox_exact"):
    d = json.loads(open("$O/%s.jsonl" % f).read().splitlines()[-1])
    print(f, "us/step %.2f frac %.3f" % (d["ms_per_step"] * 1e3, d["roofline"]["frac"]), "fused", d.get("fused_rollout", {}).get("ms_per_step"))
print(open("$O/closed_loop.jsonl").read())
PY
