#!/bin/bash
# Development tool: build a variant of the planner's translation unit with extra flags and link it with the already
# built env objects into build_ab/NAME.so (select with UAMENV_LIB=build_ab/NAME.so).
# usage: tools/build_mcts_variant.sh NAME [extra hipcc flags, e.g. -DUAM_MCTS_ABL=1]
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gym-guidance-collision-avoidance-single_amd/csrc
mkdir -p $R/build_ab
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o $R/build_ab/mcts_$NAME.o $C/uam_mcts.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_ab/$NAME.so $C/build/uamenv.o $C/build/uam_mt_kernels.o $R/build_ab/mcts_$NAME.o
rm -f $R/build_ab/mcts_$NAME.o
