O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "rollout_keeps or rollout_policy" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout -k 10 150 python tools/fuzz_step_many.py 31 60 > $O/fuzz.log 2>&1; tail -n 1 $O/fuzz.log
for s in "8192 64 64 randintruder" "2048 64 64 randintruder" "8192 32 64 randintruder"; do
  UAMENV_DEBUG=1 timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v "amdgpu.ids\|step kernel" | sed 's/.*three-wave geometry/geometry/' | tee -a $O/probe.log
  UAMENV_MANY3=off timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v amdgpu.ids | sed 's/^/two-wave: /' | tee -a $O/probe.log
done
timeout -k 10 300 python tools/bench_closed_loop.py --total 2048 2>&1 | grep '^{' | grep randintruder | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['variant'], d['envs'], d['intruders'], d['mode'], round(d['us_per_step'], 2))"
