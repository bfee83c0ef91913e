O=gpurun_out/r03c; mkdir -p $O
UAMENV_LIB=$PWD/build_ab/stamp.so timeout -k 10 120 python tools/many_probe.py 8192 32 64 continuous 6 > $O/stamp_8192x32.log 2>&1
UAMENV_LIB=$PWD/build_ab/stamp.so timeout -k 10 120 python tools/many_probe.py 4096 8 64 discrete9 6 > $O/stamp_4096x8.log 2>&1
tail -n 3 $O/stamp_8192x32.log; tail -n 3 $O/stamp_4096x8.log
