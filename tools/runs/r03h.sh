O=gpurun_out/r03h; mkdir -p $O
python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "rollout_keeps or rollout_policy" > $O/pytest1.log 2>&1 || { tail -n 30 $O/pytest1.log; exit 1; }
timeout -k 10 150 python tools/fuzz_step_many.py 14 60 > $O/fuzz.log 2>&1 || { tail -n 20 $O/fuzz.log; exit 1; }
for E in 4 8 16 32; do for s in "8192 32 256 continuous" "16384 32 128 continuous" "4096 32 256 continuous" "4096 8 256 discrete9" "2048 64 128 discrete9"; do
  echo "minE=$E" >> $O/probe.log
  UAMENV_MANY3_E=$E UAMENV_DEBUG=1 timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v "amdgpu.ids\|step kernel" | sed 's/.*three-wave geometry/geometry/' >> $O/probe.log
done; done
UAMENV_LIB=$PWD/build_ab/stamp.so timeout -k 10 120 python tools/many_probe.py 8192 32 64 continuous 6 > $O/stamp_8192x32.log 2>&1
tail -n 2 $O/pytest1.log; tail -n 1 $O/fuzz.log; cat $O/probe.log; tail -n 3 $O/stamp_8192x32.log
