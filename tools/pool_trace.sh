#!/bin/bash
# rocprofv3 --kernel-trace (+ copies) of a whole pool of workers, every worker a process under its own profiler:
#   tools/pool_trace.sh <tag> <workers> [--no-poisson]
TAG=$1; W=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pooltrace_$TAG
rm -rf $OUT; mkdir -p $OUT/barrier
cd /tmp && export TMPDIR=/tmp
for r in $(seq 0 $((W - 1))); do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/w$r -o t -- \
      python $ROOT/tools/pool_worker.py $r $W 3 $OUT/barrier "$@" > $OUT/w$r.json 2> $OUT/w$r.err &
done
wait
cat $OUT/w*.json | grep '^{' > $ROOT/gpurun_out/${TAG}_pool_traced_w$W.jsonl
python $ROOT/tools/pool_trace.py $OUT > $ROOT/gpurun_out/${TAG}_pool_trace_w$W.json 2>&1
python - <<PY
import json
rows=[json.loads(l) for l in open('$ROOT/gpurun_out/${TAG}_pool_traced_w$W.jsonl')]
wall=max(r['t1'] for r in rows)-min(r['t0'] for r in rows)
print('workers',len(rows),'pages/s',round(sum(r['pages'] for r in rows)/wall,1),'cpu ms/page',round(sum(r['cpu_s'] for r in rows)/sum(r['pages'] for r in rows)*1e3,2))
PY
du -sh $OUT | tail -1
rm -rf $OUT      # the raw traces stay on the box
