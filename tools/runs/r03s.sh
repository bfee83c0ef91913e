O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "rollout_keeps or rollout_policy" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
for s in "32768 32 64 continuous" "65536 32 32 discrete9"; do
  timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.log
  UAMENV_MANY4_GENERIC=1 timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v amdgpu.ids | sed 's/^/generic: /' | tee -a $O/probe.log
done
