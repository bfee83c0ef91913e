O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "rollout_keeps or rollout_policy" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout -k 10 150 python tools/fuzz_step_many.py 23 60 > $O/fuzz.log 2>&1; tail -n 1 $O/fuzz.log
for lib in ""; do
for b in 8192 16384; do
  UAMENV_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python bench.py --scaling strong --batch $b --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep '^{' | python -c "
import sys, json
x = json.loads(sys.stdin.read()); print('$lib', x['config']['envs_per_gpu'], round(x['ms_per_step']*1e3, 3), 'us/step', x['config']['launch'])"
done; done
