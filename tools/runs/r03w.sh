O=gpurun_out/r03w; mkdir -p $O
UAMENV_LIB=$PWD/build_ab/stamp.so timeout -k 10 120 python tools/many_probe.py 8192 32 64 continuous 8 > $O/stamp_8192x32.log 2>&1
tail -n 2 $O/stamp_8192x32.log
