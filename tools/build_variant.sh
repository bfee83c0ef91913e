#!/bin/bash
# Development tool: build a variant of libuamenv_hip.so with extra compiler flags into build_ab/NAME.so (git-ignored;
# travels to the GPU box with the snapshot); select it at run time with UAMENV_LIB=build_ab/NAME.so.
# usage: tools/build_variant.sh NAME [extra hipcc flags, e.g. -DUAM_STAMP=1]
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gym-guidance-collision-avoidance-single_amd/csrc
mkdir -p $R/build_ab/obj_$NAME
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o $R/build_ab/obj_$NAME/uamenv.o $C/uamenv.hip &
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c -o $R/build_ab/obj_$NAME/uam_mt_kernels.o $C/uam_mt_kernels.hip &
/opt/rocm/bin/hipcc $F "$@" -c -o $R/build_ab/obj_$NAME/uam_mcts.o $C/uam_mcts.hip &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_ab/$NAME.so $R/build_ab/obj_$NAME/uamenv.o $R/build_ab/obj_$NAME/uam_mt_kernels.o $R/build_ab/obj_$NAME/uam_mcts.o
rm -rf $R/build_ab/obj_$NAME
ls -la $R/build_ab/$NAME.so
