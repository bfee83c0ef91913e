O=gpurun_out/r03q; mkdir -p $O
for lib in warm0 warm64; do
echo "== $lib" | tee -a $O/mt_probe.log
UAMENV_LIB=$PWD/build_ab/$lib.so timeout -k 10 200 python tools/mt_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 4 | tee -a $O/mt_probe.log
done
