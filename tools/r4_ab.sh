#!/bin/bash
# A/B of alternative library builds on the GPU box: [BENCH_ARGS="..."] tools/r4_ab.sh <lib> [<lib> ...]; short bench at batch 256 for each
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for LIB in "$@"; do
  export VKX_LIB=$PWD/vkit_amd/$LIB
  timeout 600 python bench.py --batch 256 --steps 10 --warmup 2 --cpu-sample 0 --cpu-procs 0 --verify 2 --extra-legs 0 $BENCH_ARGS 2> gpurun_out/ab_$LIB.err > gpurun_out/ab_$LIB.json
  python - <<PY
import json
r = json.loads(open('gpurun_out/ab_$LIB.json').readline())
print('$LIB', 'Mpx/s', round(r['value']), 'ms/step', round(r['ms_per_step'], 3), json.dumps(r['roofline'].get('kernels_ms_per_step')))
PY
done
