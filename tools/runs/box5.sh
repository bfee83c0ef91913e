mkdir -p gpurun_out/box
for pre in hold "" "" hold "" churn ""; do
  PROBE_PRE=$pre timeout -k 10 200 python tools/box_probe.py 2>/dev/null | grep '^{' | tee -a gpurun_out/box/pre2_$(date +%s).jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(repr(d['pre']), d['gpu'][1], round(d['step_ms_1M_envs'],4), round(d['step_frac_of_8TBps'],3), 'copy', round(d['copy_frac_of_8TBps'],3))"
done
