O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "mt19937 or mt or replay or seed or shared or golden or reference or lockstep" > $O/pytest_mt.log 2>&1; tail -n 3 $O/pytest_mt.log
for lib in "" warm0 warm1k; do
echo "== lib=$lib" | tee -a $O/mt_probe.log
UAMENV_LIB=${lib:+$PWD/build_ab/$lib.so} timeout -k 10 200 python tools/mt_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 4 | tee -a $O/mt_probe.log
done
