mkdir -p gpurun_out/box
for i in 1 2 3; do
  timeout -k 10 120 python tools/box_probe.py 2>/dev/null | grep '^{' | sed 's/^/arena: /' | tee -a gpurun_out/box/arena_$(date +%s).jsonl | cut -c1-330
  UAMENV_NO_ARENA=1 timeout -k 10 120 python tools/box_probe.py 2>/dev/null | grep '^{' | sed 's/^/separate: /' | tee -a gpurun_out/box/arena_$(date +%s).jsonl | cut -c1-330
done
