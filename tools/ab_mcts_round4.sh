#!/bin/bash
# Development probe: the planner, the committed build (build_ab/prev.so) against the working tree, same box
for rep in 1 2; do
for lib in build_ab/prev.so ""; do
  for s in "--batch 8192 --intruders 64" "--batch 2048 --intruders 20" "--batch 8192 --intruders 32" ; do
    echo "lib=${lib:-current} $(UAMENV_LIB=$lib python3 tools/bench_mcts.py --reps 5 --cpu-roots 0 $s 2>/dev/null | grep '^{' | cut -c1-160)"
  done
done
done
