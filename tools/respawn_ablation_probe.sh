for lib in "" build_ab/noresp.so; do
  for s in "8192 32 256 continuous" "16384 32 128 continuous" "4096 8 256 discrete9" "2048 32 256 continuous"; do
    echo "lib=$lib $(UAMENV_LIB=$lib python3 tools/many_probe.py $s 20 2>&1 | grep -v amdgpu)"
  done
done
UAMENV_LIB=build_ab/stamp.so python3 tools/many_probe.py 8192 32 256 continuous 4 2>&1 | grep -v amdgpu | tail -n 4
