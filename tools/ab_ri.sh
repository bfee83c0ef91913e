#!/bin/bash
# Development tool (GPU box): RandIntruder step time for a list of library variants (build_ab/NAME.so; "-" = shipped).
OUT=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset UAMENV_LIB; else export UAMENV_LIB=$PWD/build_ab/$v.so; fi
  echo "== $v" >> $OUT
  UAMENV_MANY=off timeout -k 10 100 python tools/many_probe.py 8192 64 64 randintruder 40 2>&1 | grep us/step >> $OUT
  UAMENV_MANY=off timeout -k 10 100 python tools/many_probe.py 65536 32 32 randintruder 40 2>&1 | grep us/step >> $OUT
  timeout -k 10 100 python tools/many_probe.py 8192 64 64 randintruder 40 2>&1 | grep us/step >> $OUT
done
cat $OUT
