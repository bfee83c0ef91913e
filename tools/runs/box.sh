mkdir -p gpurun_out/box
timeout -k 10 120 python tools/box_probe.py 2>/dev/null | grep '^{' | tee -a gpurun_out/box/probe_$(date +%s).jsonl
