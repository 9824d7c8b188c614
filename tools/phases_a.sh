#!/bin/bash
# time of the fused kernel cut inside phase A (VKX_FUSED_PHASES 11: after the chunk load, 12: after the work-list scan,
# 13: after the span raster, 1: whole phase A), batch 64
cd /root/repo
for P in 11 1; do
  echo "== phases $P"
  VKX_FUSED_PHASES=$P timeout 300 python bench.py --batch 64 --steps 5 --warmup 2 --cpu-sample 0 --verify 0 --cpu-procs 0 --noise-workers 32 --extra-legs 0 2> gpurun_out/ph.err | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['roofline']['kernels_ms_per_step'])"
done
