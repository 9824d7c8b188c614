#!/bin/bash
# Collects rocprofv3 PMC counters for the bench (run on the GPU box via gpurun). Usage: tools/pmc.sh <tag> <batch>
# Each --pmc group is its own pass, with --kernel-trace only (no sys/hip/hsa tracing).
TAG=${1:-r1}; BATCH=${2:-32}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (VKX_CHAIN_CHUNKS=1: one k_np_draw / k_chain_fused launch per batch, so that "per launch" is "per batch" in tools/pmc_parse.py; the
#  counters per image do not depend on how the library cuts the batch into chunks)
export VKX_CHAIN_CHUNKS=1
BENCH="python $ROOT/bench.py --batch $BATCH --steps 2 --warmup 1 --cpu-sample 0 --verify 0 --extra-legs 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $BENCH > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
find $OUT -name "*counter_collection.csv" | head
