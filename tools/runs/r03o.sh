O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "rollout_keeps" > $O/pytest1.log 2>&1; tail -n 3 $O/pytest1.log
timeout -k 10 120 python tools/bench_copy.py 2>&1 | grep -v amdgpu.ids | tee $O/copy_shapes.log
