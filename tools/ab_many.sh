#!/bin/bash
# Development tool (GPU box): time the looped kernel and the headline step kernel for a list of "ENVVARS" settings.
# usage: tools/ab_many.sh OUTFILE "VAR=val VAR2=val" "..." ...
OUT=$1; shift
for setting in "$@"; do
  echo "== $setting" >> $OUT
  for s in "8192 32 64 continuous" "4096 8 64 discrete9" "16384 32 64 continuous" "8192 64 64 randintruder"; do
    env $setting timeout -k 10 60 python tools/many_probe.py $s 100 2>&1 | grep -v amdgpu.ids >> $OUT
  done
  env $setting UAMENV_MANY=off timeout -k 10 60 python tools/many_probe.py 65536 32 32 discrete9 60 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
