O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/pytest_all.log
tail -n 5 $O/pytest_all.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 $O/smoke.log
for b in 8192 16384 32768; do
  timeout -k 10 120 python bench.py --scaling strong --batch $b --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' >> $O/strong_shard_shapes.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r03m/strong_shard_shapes.jsonl"):
    d = json.loads(l); print(d["config"]["envs_per_gpu"], round(d["ms_per_step"]*1e3, 3), "us/step", d["config"]["launch"], d["config"]["chunk"])
PY
