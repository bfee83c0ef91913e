#!/bin/bash
# Development probe (second look): size or concurrency?  Half batches alone, two half batches taking turns on one stream,
# larger batches split 1 / 2 / 4 ways.
R=${GRAFT_REPO_ROOT:-$PWD}
export PROBE_REPS=6
python3 $R/tools/split_stream_probe.py 524288 32 16 1
python3 $R/tools/split_stream_probe.py 262144 32 16 1
PROBE_SAME_STREAM=1 python3 $R/tools/split_stream_probe.py 1048576 32 16 2
PROBE_SAME_STREAM=1 python3 $R/tools/split_stream_probe.py 1048576 32 16 4
python3 $R/tools/split_stream_probe.py 2097152 32 16 1 2 4
python3 $R/tools/split_stream_probe.py 262144 32 32 1 2
python3 $R/tools/split_stream_probe.py 131072 32 32 1 2
