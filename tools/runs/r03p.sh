O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -n 3 $O/pytest_all.log
timeout -k 10 200 python tools/mt_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/mt_probe.log
timeout -k 10 200 python tools/mt_probe.py 65536 32 goal 2>&1 | grep -v amdgpu.ids | tee $O/mt_probe_goal.log
timeout -k 10 200 python tools/bench_variants.py randintruder 2>&1 | grep -v amdgpu.ids | tee $O/ri_65536.log
timeout -k 10 100 python tools/many_probe.py 8192 64 64 randintruder 30 2>&1 | grep -v amdgpu.ids | tee $O/ri_8192x64_many.log
UAMENV_MANY=off timeout -k 10 100 python tools/many_probe.py 8192 64 64 randintruder 30 2>&1 | grep -v amdgpu.ids | tee $O/ri_8192x64_single.log
