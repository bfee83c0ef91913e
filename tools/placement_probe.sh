#!/bin/bash
# Development probe: is the 1 M-env step's bimodal speed (0.61 / 0.73 of peak from process to process) tied to where the
# arrays land (virtual addresses printed by UAMENV_DEBUG), and does a batch split over sub-batch handles avoid it?
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-placement}; mkdir -p $O
for i in 1 2 3 4 5 6; do
  UAMENV_DEBUG=1 PROBE_REPS=6 python3 $R/tools/split_stream_probe.py 1048576 32 16 1 > $O/s1_$i.out 2> $O/s1_$i.err
  grep -h "us_per_step" $O/s1_$i.out; grep -h "arrays" $O/s1_$i.err
done
for i in 1 2 3 4; do
  UAMENV_DEBUG=1 PROBE_REPS=6 python3 $R/tools/split_stream_probe.py 1048576 32 16 2 > $O/s2_$i.out 2> $O/s2_$i.err
  grep -h "us_per_step" $O/s2_$i.out; grep -h "arrays" $O/s2_$i.err
done
