#!/bin/bash
# round 4 quick check on the GPU box: numpy-stream parity tests + a short bench at batch 256 (kernel times per step)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-q}
timeout 1200 python -m pytest tests/test_gpu_np_stream.py tests/test_gpu_hostpipe.py tests/test_gpu_parity.py -x -q -k "${2:-not nothing}" > gpurun_out/${TAG}_tests.log 2>&1; tail -15 gpurun_out/${TAG}_tests.log
timeout 600 python bench.py --batch 256 --steps 10 --warmup 2 --cpu-sample 0 --cpu-procs 0 --verify 2 --extra-legs 0 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
r = json.loads(open('gpurun_out/${TAG}_bench.json').readline())
print('Mpx/s', round(r['value']), 'ms/step', round(r['ms_per_step'], 3))
print(json.dumps(r['roofline'].get('kernels_ms_per_step')))
print(json.dumps(r.get('noise_stream', {}).get('kernels_ms_per_step')))
PY
