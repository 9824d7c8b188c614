#!/bin/bash
# timing A/B between library builds: tools/ab2.sh <libA> <libB> ...
cd /root/repo
for L in "$@"; do
  echo "== $L"
  VKX_LIB=/root/repo/vkit_amd/$L timeout 300 python bench.py --batch 64 --steps 5 --warmup 2 --cpu-sample 1 --cpu-procs 0 --noise-workers 32 --extra-legs 0 2> gpurun_out/ab2.err | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('Mpx/s', round(r['value']), 'ms/step', round(r['ms_per_step'],3), r['roofline']['kernels_ms_per_step'])"
done
