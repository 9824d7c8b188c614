#!/bin/bash
# quick A/B on the GPU box: chain parity tests + a short bench (batch 64)
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chain" > gpurun_out/quick_tests.log 2>&1; tail -3 gpurun_out/quick_tests.log
timeout 300 python bench.py --batch 64 --steps 5 --warmup 2 --cpu-sample 1 --cpu-procs 0 --noise-workers 32 --extra-legs 0 2> gpurun_out/quick_bench.err | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('Mpx/s', round(r['value']), 'ms/step', round(r['ms_per_step'],3), r['roofline']['kernels_ms_per_step'])"
