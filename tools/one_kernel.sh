#!/bin/bash
# Compile only the headline step kernel (and with -DWITH_MANY the looped kernel) from a given csrc directory and print
# its register usage: seconds instead of the 70 s full build.  usage: tools/one_kernel.sh [csrc-dir] [extra hipcc flags]
DIR=${1:-gym-guidance-collision-avoidance-single_amd/csrc}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -I"$DIR" -c /tmp/k/one.hip -o /tmp/k/one.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: //' | paste - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
