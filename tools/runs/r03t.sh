O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for b in 8192 16384 32768; do
  timeout -k 10 120 python bench.py --scaling strong --batch $b --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' >> $O/strong_shard_shapes.jsonl
done
timeout -k 10 300 python bench.py --rng mt19937 --precision exact --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' > $O/bench_mt19937_exact.jsonl
timeout -k 10 300 python bench.py --precision exact --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' >> $O/bench_mt19937_exact.jsonl
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --same-device --steps 64 --warmup 16 --reps 2 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' > $O/bench_n2_gloo_same_device_rehearsal.json
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03t/bench_n1.json") if l.startswith("{")][-1])
r = d["roofline"]; h = r["hbm_resident"]
print("value %.3f G  ms/step %.4f  frac %.3f  sustained %s" % (d["value"]/1e9, d["ms_per_step"], r["frac"], d["sustained"]))
print("copy own %.4f torch %.4f frac_of_copy %.3f" % (r["copy_same_bytes_ms"], r["torch_copy_same_bytes_ms"], r["frac_of_copy"]))
print("resident: %.4f ms frac %.3f kernel_frac %s copy %.4f (%.3f of peak) torch %.4f frac_of_copy %.3f" % (h["avg_launch_ms"], h["frac"], h.get("kernel_frac"), h["copy_same_bytes_ms"], h["copy_frac_of_peak"], h["torch_copy_same_bytes_ms"], h["frac_of_copy"]))
print("strong N=1: %.3f us %s | fused %.3f us %s" % (d["strong"]["ms_per_step"]*1e3, d["strong"]["kernel"], d["fused_rollout"]["ms_per_step"]*1e3, d["fused_rollout"]["kernel"]))
print("cpu", json.dumps(d["cpu_baseline"])[:700])
for l in open("gpurun_out/r03t/strong_shard_shapes.jsonl"):
    x = json.loads(l); print("shard", x["config"]["envs_per_gpu"], round(x["ms_per_step"]*1e3, 3), "us/step", x["config"]["launch"], x["config"]["chunk"], "value %.2f G" % (x["value"]/1e9))
for l in open("gpurun_out/r03t/bench_mt19937_exact.jsonl"):
    x = json.loads(l); print(x["config"]["rng"], x["config"]["positions"], "value ms/step %.4f frac %.3f | fused %.4f frac %.3f" % (x["ms_per_step"], x["roofline"]["frac"], x["fused_rollout"]["ms_per_step"], x["fused_rollout"]["roofline_frac"]))
PY
