set -x
mkdir -p gpurun_out/r03a
python -m pytest tests/test_gpu_rccl.py tests/test_gpu_rollout.py -x -q -m gpu > gpurun_out/r03a/pytest1.log 2>&1 && \
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "continuous2 or packaged or drop_in" > gpurun_out/r03a/pytest2.log 2>&1 && \
UAMENV_LIB=$PWD/build_ab/stamp.so timeout -k 10 120 python tools/many_probe.py 8192 32 64 continuous 10 > gpurun_out/r03a/stamp_8192x32.log 2>&1 && \
timeout -k 10 300 python bench.py > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err && \
timeout -k 10 120 python bench.py --scaling strong --batch 8192 --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r03a/strong_8192.json 2>&1
echo rc=$?
tail -3 gpurun_out/r03a/pytest1.log gpurun_out/r03a/pytest2.log
