#!/bin/bash
# Round record on the GPU box: PMC passes (batch 32) first -- their summary is what the bench line's roofline.traffic / valu_issue_frac
# read, tied to the kernel sources by a digest --, then the default bench line, then rocprofv3 kernel stats at batch 64 and 256.
# Usage (through gpurun): ./tools/record.sh <tag>     (afterwards, here: tools/pmc_parse.py <tag> 32 regenerates profiles/<tag>_*)
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
python -c "import bench; print(bench.kernel_source_digest())" > gpurun_out/digest_$TAG.txt
$ROOT/tools/pmc.sh $TAG 32
python $ROOT/tools/pmc_parse.py $TAG 32 0 > /dev/null          # writes profiles/current_traffic.json on this box for the bench below
cd $ROOT
unset VKX_CHAIN_CHUNKS
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/stats_$TAG -o stats -- \
    python $ROOT/bench.py --batch 64 --steps 3 --warmup 1 --cpu-sample 0 --verify 0 --extra-legs 0 \
    > $ROOT/gpurun_out/stats_$TAG.log 2>&1; echo "stats rc=$?"
# the same summary at the bench's own batch size (256 images per launch): roofline.frac is reproducible from profiles/ alone
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/stats256_$TAG -o stats -- \
    python $ROOT/bench.py --batch 256 --steps 5 --warmup 1 --cpu-sample 0 --verify 0 --extra-legs 0 \
    > $ROOT/gpurun_out/stats256_$TAG.log 2>&1; echo "stats256 rc=$?"
# the timeline of one step (which kernel runs under which): rocprofv3 --kernel-trace of a short run, tools/timeline.py
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tl_$TAG -o tl -- \
    python $ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 --verify 0 --extra-legs 0 > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/tl_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && VKX_TL_CAMERA=1 python $ROOT/tools/timeline.py $f 2 > $ROOT/gpurun_out/${TAG}_timeline.txt
rm -rf $ROOT/gpurun_out/tl_$TAG
