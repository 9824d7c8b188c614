#!/bin/bash
# Builds vkit_amd/libvkx_np_<tag>.so: the library with nprand.hip compiled under extra flags (A/B experiments on the numpy
# stream kernels; run with VKX_LIB=vkit_amd/libvkx_np_<tag>.so).  Usage: tools/np_alt.sh <tag> [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
C=$ROOT/vkit_amd/csrc
mkdir -p $C/_alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function "$@" -c $C/nprand.hip -o $C/_alt/nprand_$TAG.o
OBJS=$(ls $C/_build/*.o | grep -v nprand.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/vkit_amd/libvkx_np_$TAG.so $OBJS $C/_alt/nprand_$TAG.o
echo built $ROOT/vkit_amd/libvkx_np_$TAG.so
