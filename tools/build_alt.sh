#!/bin/bash
# Builds vkit_amd/libvkx_alt.so with an alternative fused.hip (A/B experiments: VKX_LIB=.../libvkx_alt.so).
# Usage: tools/build_alt.sh <fused source> [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$(readlink -f "$1"); shift
C=$ROOT/vkit_amd/csrc
mkdir -p $C/_alt
cp "$SRC" $C/_alt_fused.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function "$@" -c $C/_alt_fused.hip -o $C/_alt/fused.o
rm -f $C/_alt_fused.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/vkit_amd/libvkx_alt.so $C/_build/ctx.o $C/_build/remap.o $C/_build/grid.o $C/_build/mls.o $C/_build/photo.o $C/_build/composite.o $C/_build/polygon.o $C/_build/resize.o $C/_alt/fused.o $C/_build/chain.o $C/_build/host_api.o
echo built $ROOT/vkit_amd/libvkx_alt.so
