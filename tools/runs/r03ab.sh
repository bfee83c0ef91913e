O=gpurun_out/r03ab; mkdir -p $O
cat /sys/fs/cgroup/cpu.max > $O/cpu_max.txt 2>&1
( time timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/time.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03ab/bench_n1.json") if l.startswith("{")][-1])
print("value %.3f G" % (d["value"]/1e9), "cpu:", json.dumps(d["cpu_baseline"])[:900])
PY
cat $O/time.txt $O/cpu_max.txt
