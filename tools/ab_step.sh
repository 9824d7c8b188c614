#!/bin/bash
# A/B of the C3 step under environment switches: tools/ab_step.sh "VAR=1 OTHER=2" "VAR=0" ...   (one bench process per setting)
for cfg in "$@"; do
  env $cfg python bench.py --extra-legs 0 --cpu-sample 0 --steps ${STEPS:-40} --warmup 5 --verify ${VERIFY:-2} 2>&1 | tail -1 | python -c "
import json,sys
cfg=sys.argv[1]
try:
    d=json.loads(sys.stdin.read())
    print(cfg, '| ms_per_step %.3f' % d['ms_per_step'], '| kernels', d['roofline']['kernels_ms_per_step'], '| verified', d['config']['verified_against_oracle'])
except Exception as e:
    print(cfg, 'FAILED', e)
" "$cfg"
done
