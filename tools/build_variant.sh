#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...  → opencl-raytracing_amd/variants/<name>.so
ROOT=$(dirname $(dirname $(readlink -f $0)))
NAME=$1; shift
mkdir -p $ROOT/opencl-raytracing_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -fPIC -shared -Wall -Wno-unused-function "$@" $ROOT/opencl-raytracing_amd/csrc/rt_amd.hip -o $ROOT/opencl-raytracing_amd/variants/$NAME.so
