O=gpurun_out/r03ac; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packaged or bad_arguments" > $O/pytest.log 2>&1; tail -n 25 $O/pytest.log
