#!/bin/bash
# rocprofv3 --kernel-trace --stats of the single-stage kernels at 8192^2 and of the C2 / C5 configurations on the current code
# (verdict item: "re-take rocprofv3 --stats for tools/kernels.py 8192, c2.py, c5.py on the final code").  Usage (through gpurun): tools/retake_stats.sh <tag>
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for what in "kernels:tools/kernels.py 8192" "c2:tools/c2.py" "c5:tools/c5.py"; do
  name=${what%%:*}; cmd=${what#*:}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/stats_${name}_$TAG -o stats -- python $ROOT/$cmd > $ROOT/gpurun_out/stats_${name}_$TAG.log 2>&1
  echo "$name rc=$?"
  f=$(find $ROOT/gpurun_out/stats_${name}_$TAG -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $ROOT/gpurun_out/${TAG}_${name}_kernel_stats.csv
done
cd $ROOT && python tools/kernels.py 8192 2>/dev/null | tail -1 > gpurun_out/${TAG}_single_stage_kernels.json
