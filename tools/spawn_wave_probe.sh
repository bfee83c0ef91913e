#!/bin/bash
# Development probe: the looped kernel's spawn wave at s_setprio 0 / 2 / 3 (build_ab/spN.so) and switched off
for lib in sp0 sp2 sp3 sp0 sp2 sp3; do
  for s in "8192 32 256 continuous" "2048 32 256 continuous" "4096 8 256 discrete9"; do
    echo "$lib $(UAMENV_LIB=build_ab/$lib.so python3 tools/many_probe.py $s 20 2>&1 | grep -v amdgpu)"
  done
done
for s in "8192 32 256 continuous" "2048 32 256 continuous" "4096 8 256 discrete9"; do
  echo "off $(UAMENV_SPAWN_WAVE=0 UAMENV_LIB=build_ab/sp2.so python3 tools/many_probe.py $s 20 2>&1 | grep -v amdgpu)"
done
for sw in 1 0; do UAMENV_SPAWN_WAVE=$sw UAMENV_LIB=build_ab/stamp.so python3 tools/many_probe.py 8192 32 256 continuous 3 2>&1 | grep -v amdgpu | tail -n 2; done
