O=gpurun_out/r03i; mkdir -p $O
for lib in "" build_ab/prio0.so build_ab/prio1.so; do for s in "8192 32 256 continuous" "4096 8 256 discrete9"; do
  echo "lib=$lib" >> $O/probe.log
  UAMENV_LIB=${lib:+$PWD/$lib} timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v "amdgpu.ids" >> $O/probe.log
done; done
UAMENV_LIB=$PWD/build_ab/stamp0.so timeout -k 10 120 python tools/many_probe.py 8192 32 64 continuous 6 > $O/stamp0_8192x32.log 2>&1
cat $O/probe.log; tail -n 2 $O/stamp0_8192x32.log
