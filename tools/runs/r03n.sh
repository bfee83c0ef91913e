O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "rollout_keeps" > $O/pytest1.log 2>&1; tail -n 3 $O/pytest1.log
rocm-smi --showclocks --showperflevel --showpower > $O/smi_before.txt 2>&1
for i in 1 2; do
  timeout -k 10 120 python tools/big_batch_probe.py >> $O/big_default.log 2>&1
  UAMENV_NO_NOISE_OVERLAP=1 timeout -k 10 120 python tools/big_batch_probe.py >> $O/big_no_overlap.log 2>&1
done
rocm-smi --showclocks --showperflevel --showpower > $O/smi_after.txt 2>&1
timeout -k 10 300 python bench.py --no-cpu-baseline --sustain-seconds 0 > $O/bench.json 2> $O/bench.err
grep -v amdgpu.ids $O/big_default.log; echo ---; grep -v amdgpu.ids $O/big_no_overlap.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03n/bench.json") if l.startswith("{")][-1])
r = d["roofline"]; h = r["hbm_resident"]
print("value", d["value"]/1e9, "frac", r["frac"], "copy ms own/torch", r["copy_same_bytes_ms"], r["torch_copy_same_bytes_ms"], "frac_of_copy", r["frac_of_copy"])
print("resident", h["avg_launch_ms"], h["frac"], h.get("kernel_frac"), "copy own/torch", h["copy_same_bytes_ms"], h["torch_copy_same_bytes_ms"], "copy frac of peak", h["copy_frac_of_peak"], "frac_of_copy", h["frac_of_copy"])
print("strong", d["strong"]["ms_per_step"]*1e3, d["strong"]["kernel"], "fused", d["fused_rollout"]["ms_per_step"]*1e3, d["fused_rollout"]["kernel"])
PY
grep -i "sclk\|mclk\|perf" $O/smi_before.txt | head -8
