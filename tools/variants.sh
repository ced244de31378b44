#!/bin/bash
# Kernel experiments: build libdagnet_hip_<tag>.so variants that differ only in the flags of the split-bf16 rollout object
# (DAG-Net L2/Z32 instantiation only: -DDN_V2_ONLY), reusing every other object of the regular build.
#   tools/variants.sh <tag> "<extra hipcc flags>" [<tag> "<flags>" ...]          (run here, on the CPU box)
# Run one with:  DAGNET_HIP_LIB=dagnet_amd/_lib/libdagnet_hip_<tag>.so python bench.py --no-cpu-baseline --no-reference-loop
set -e
cd "$(dirname "$0")/../dagnet_amd/csrc"
LIB=../_lib
OTHERS=$(ls $LIB/obj/*.o | grep -v rollout_v2_bf3.o)
mkdir -p $LIB/var
while [ $# -ge 2 ]; do
  TAG=$1; FLAGS=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -fno-honor-nans -DDN_V2_BF=1 -DDN_V2_ONLY -fno-slp-vectorize -DDN_V2_NO_SLP $FLAGS -c -o $LIB/var/$TAG.o rollout_v2.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $LIB/libdagnet_hip_$TAG.so $OTHERS $LIB/var/$TAG.o && echo built $TAG ) &
done
wait
