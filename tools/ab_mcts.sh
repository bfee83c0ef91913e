#!/bin/bash
# Development tool (GPU box): time the batched planner for a list of library variants (build_ab/NAME.so).
# usage: tools/ab_mcts.sh OUTFILE NAME [NAME ...]     (NAME "-" = the shipped library)
OUT=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset UAMENV_LIB; else export UAMENV_LIB=$PWD/build_ab/$v.so; fi
  for shape in "8192 64" "2048 20"; do
    set -- $shape
    echo -n "$v $shape " >> $OUT
    timeout -k 10 120 python tools/bench_mcts.py --batch $1 --intruders $2 --cpu-roots 0 --model ${MCTS_MODEL:-randintru} 2>&1 | grep gpu_ms | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%.3f ms' % d['gpu_ms_per_batched_decision'])" >> $OUT
  done
done
cat $OUT
