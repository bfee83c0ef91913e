mkdir -p gpurun_out/box
for cfg in "none" "0" "4352" "69888" "1052928" "none" "2101504"; do
  for rep in 1 2; do
  if [ "$cfg" = "none" ]; then E=""; else E="UAMENV_ARENA=1 UAMENV_ARENA_PAD=$cfg"; fi
  env $E timeout -k 10 200 python tools/box_probe.py 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pad=$cfg', d['gpu'][1], round(d['step_ms_1M_envs'],4), round(d['step_frac_of_8TBps'],3), 'copy', round(d['copy_frac_of_8TBps'],3))" | tee -a gpurun_out/box/pads.log
  done
done
