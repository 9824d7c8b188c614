#!/bin/bash
# A/B of environment switches on the GPU box: tools/r4_env.sh "VAR=val VAR2=val" ... ; short bench at batch 256 for each
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
i=0
for E in "$@"; do
  i=$((i+1))
  env $E timeout 600 python bench.py --batch 256 --steps 10 --warmup 2 --cpu-sample 0 --cpu-procs 0 --verify 0 --extra-legs 0 2> gpurun_out/env_$i.err > gpurun_out/env_$i.json
  python - <<PY
import json
r = json.loads(open('gpurun_out/env_$i.json').readline())
print('$E', '| Mpx/s', round(r['value']), 'ms/step', round(r['ms_per_step'], 3), json.dumps(r['roofline'].get('kernels_ms_per_step')))
PY
done
