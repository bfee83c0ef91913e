#!/bin/bash
# Development tool: throughput of the BASELINE.json configurations and a few modes on one GPU.
R=${GRAFT_REPO_ROOT:-$PWD}
run() { python $R/bench.py --no-cpu-baseline --steps ${STEPS:-1000} --warmup 100 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-58s %8.3f G env-steps/s %7.2f us/step  step-roofline %.3f  sweep %.2f us own %.2f us' % (' '.join(sys.argv[1:]), d['value']/1e9, d['ms_per_step']*1e3, r['whole_step']['frac'], r['avg_launch_ms']*1e3, r['own_kernel_avg_ms']*1e3))" "$@"; }
run --batch 4096 --intruders 8
run --batch 65536 --intruders 32
run --batch 65536 --intruders 32 --policy goal
run --batch 1048576 --intruders 32
run --batch 65536 --intruders 32 --variant continuous
run --batch 8192 --intruders 64 --variant randintruder
run --batch 65536 --intruders 80
run --batch 65536 --intruders 32 --precision exact
STEPS=200 run --batch 65536 --intruders 32 --rng mt19937 --precision exact
run --batch 262144 --intruders 8
