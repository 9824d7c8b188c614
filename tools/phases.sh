#!/bin/bash
# time of the fused kernel cut after phase A / after C+D (VKX_FUSED_PHASES), batch 64
cd /root/repo
for P in 1 2 0; do
  echo "== phases $P"
  VKX_FUSED_PHASES=$P timeout 300 python bench.py --batch 64 --steps 5 --warmup 2 --cpu-sample 0 --verify 0 --cpu-procs 0 --noise-workers 32 --extra-legs 0 2> gpurun_out/ph.err | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['roofline']['kernels_ms_per_step'])"
done
