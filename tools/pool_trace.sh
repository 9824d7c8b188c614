#!/bin/bash
# rocprofv3 --kernel-trace (+ copies) of a whole worker pool: tools/pool_trace.sh <tag> <workers> [extra pool_scale args]
TAG=$1; W=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pooltrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- \
    python $ROOT/tools/pool_scale.py --workers $W --seconds 3 --modes pipeline "$@" > $OUT/pool.json 2> $OUT/pool.err
echo "trace rc=$?"
python $ROOT/tools/pool_trace.py $OUT > $ROOT/gpurun_out/${TAG}_pool_trace_w$W.json 2>&1
cp $OUT/pool.json $ROOT/gpurun_out/${TAG}_pool_traced_w$W.json
du -sh $OUT | tail -1
find $OUT -name "*.csv" -size +20M -delete     # the raw traces stay on the box unless small
