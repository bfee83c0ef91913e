O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py -x -q -m gpu > $O/pytest_graph.log 2>&1; tail -n 3 $O/pytest_graph.log
timeout -k 10 300 python tools/bench_closed_loop.py --total 2048 2>&1 | grep '^{' | tee $O/closed_loop.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['variant'], d['envs'], d['intruders'], d['mode'], round(d['us_per_step'], 2))"
UAMENV_INLINE_NOISE=0 timeout -k 10 300 python tools/bench_closed_loop.py --total 2048 2>&1 | grep '^{' | grep '"graph"' | tee $O/closed_loop_noinline.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('separate noise launch:', d['variant'], d['envs'], d['intruders'], d['mode'], round(d['us_per_step'], 2))"
timeout -k 10 120 python bench.py --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('headline', d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'])"
