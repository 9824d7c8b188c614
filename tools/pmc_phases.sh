#!/bin/bash
# VALU / SALU / LDS instructions per wavefront of k_chain_fused cut after each phase (VKX_FUSED_PHASES), batch 32.
# Usage: tools/pmc_phases.sh "<phase list>"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for P in ${1:-11 12 13 1 2 0}; do
  OUT=$ROOT/gpurun_out/pmc_ph$P
  rm -rf $OUT; mkdir -p $OUT
  VKX_FUSED_PHASES=$P timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- \
    python $ROOT/bench.py --batch 32 --steps 2 --warmup 1 --cpu-sample 0 --verify 0 --noise-workers 0 --extra-legs 0 > $OUT/p1.log 2>&1
  echo "== phases $P"; python $ROOT/tools/pmc_sum.py ph$P | grep "SQ_INSTS\|ns_p1"
done
