#!/bin/bash
# Development probe: the looped kernel with its spawn wave on (default) and off (UAMENV_SPAWN_WAVE=0)
for sw in 1 0 1 0; do
  for s in "8192 32 256 continuous" "2048 32 256 continuous" "4096 8 256 discrete9" "4096 32 256 continuous"; do
    echo "UAMENV_SPAWN_WAVE=$sw $(UAMENV_SPAWN_WAVE=$sw python3 tools/many_probe.py $s 20 2>&1 | grep -v amdgpu)"
  done
done
for sw in 1 0; do UAMENV_SPAWN_WAVE=$sw UAMENV_LIB=build_ab/stamp.so python3 tools/many_probe.py 8192 32 256 continuous 3 2>&1 | grep -v amdgpu | tail -n 2; done
