#!/bin/bash
# Development tool: throughput of the BASELINE.json configurations and a few other shapes / modes on one GPU
# (one launch per step unless the line says "fused": chunks of 32 steps from an action tape in one launch).
R=${GRAFT_REPO_ROOT:-$PWD}
run() { python3 $R/bench.py --no-build --no-cpu-baseline --no-secondary --sustain-seconds 0 --steps ${STEPS:-1000} --warmup 100 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-66s %8.3f G env-steps/s %8.2f us/step  frac %.3f  %s' % (' '.join(sys.argv[1:]), d['value']/1e9, d['ms_per_step']*1e3, r['frac'], r['kernel']))" "$@"; }
run --batch 4096 --intruders 8
run --batch 65536 --intruders 32
run --batch 65536 --intruders 32 --policy goal
run --batch 1048576 --intruders 32
run --batch 65536 --intruders 32 --variant continuous
run --batch 8192 --intruders 64 --variant randintruder
run --batch 65536 --intruders 32 --variant randintruder
run --batch 65536 --intruders 80
run --batch 65536 --intruders 16
run --batch 65536 --intruders 64
run --batch 262144 --intruders 8
run --batch 65536 --intruders 32 --variant her9
run --batch 65536 --intruders 32 --variant mcts
run --batch 65536 --intruders 32 --precision exact
STEPS=400 run --batch 65536 --intruders 32 --rng mt19937 --precision exact
run --batch 65536 --intruders 32 --scaling strong
run --batch 8192 --intruders 32 --scaling strong
run --batch 8192 --intruders 64 --variant randintruder --scaling strong
