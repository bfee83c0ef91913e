#!/bin/bash
# Per-round auxiliary records (GPU box), written under gpurun_out/<tag>/ — copy what should be judged into profiles/:
#   bench_n1.json                      the default `python bench.py` line (N = 1, with cpu_baseline)
#   strong_shard_shapes.jsonl          BASELINE config 4's per-rank shapes on one GPU (looped kernel, no exchange)
#   closed_loop.jsonl                  tools/bench_closed_loop.py: eager / HIP graph / in-kernel policy
#   bench_n2_gloo_same_device.json     2-rank rehearsal of bench.py's own exchange path on one GPU
#   bench_mt19937_exact.jsonl          exact-position modes (MT19937, then Philox): `value` = single-step launches, `fused_rollout` = chunks
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench n1 done"
: > $OUT/strong_shard_shapes.jsonl
for b in 8192 16384 32768; do
  python3 bench.py --scaling strong --variant continuous --batch $b --steps 2048 --warmup 256 --no-secondary --no-cpu-baseline 2>> $OUT/strong.err | grep '^{' >> $OUT/strong_shard_shapes.jsonl
done
echo "strong shapes done"
python3 tools/bench_closed_loop.py 2> $OUT/closed_loop.err | grep '^{' > $OUT/closed_loop.jsonl
echo "closed loop done"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --same-device --no-cpu-baseline 2> $OUT/n2.err | grep '^{' > $OUT/bench_n2_gloo_same_device.json
echo "n2 rehearsal done"
: > $OUT/bench_mt19937_exact.jsonl
# value = one launch per step, fused_rollout = one launch per chunk, in the same line; MT19937 first, then Philox
python3 bench.py --rng mt19937 --precision exact --steps 640 --warmup 128 --reps 3 --no-cpu-baseline 2>> $OUT/mt.err | grep '^{' >> $OUT/bench_mt19937_exact.jsonl
python3 bench.py --precision exact --steps 640 --warmup 128 --reps 3 --no-cpu-baseline 2>> $OUT/mt.err | grep '^{' >> $OUT/bench_mt19937_exact.jsonl
echo "mt done"
