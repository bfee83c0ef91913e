#!/bin/bash
# Development tool (GPU box): MT19937 / exact at 65536 x 32 for a list of library variants (build_ab/NAME.so; "-" = shipped):
# single-step launches (tools/mt_probe.py, UAMENV_MANY=off) and K-step rollout chunks (tools/many_probe.py).
OUT=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset UAMENV_LIB; else export UAMENV_LIB=$PWD/build_ab/$v.so; fi
  echo "== $v" >> $OUT
  UAMENV_MANY=off timeout -k 10 100 python tools/mt_probe.py 2>&1 | grep "steady" | tail -2 >> $OUT
  PROBE_RNG=mt19937 PROBE_PRECISION=exact timeout -k 10 100 python tools/many_probe.py 65536 32 32 discrete9 20 2>&1 | grep us/step >> $OUT
done
cat $OUT
