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
OBJS=$(ls $C/_build/*.o | grep -v fused.o); /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/vkit_amd/libvkx_${ALT_TAG:-alt}.so $OBJS $C/_alt/fused.o
echo built $ROOT/vkit_amd/libvkx_${ALT_TAG:-alt}.so
