#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_chain_errors.py tests/test_gpu_camera_states.py tests/test_gpu_composite.py -x -q -m gpu > gpurun_out/r6b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r6b_tests.log
timeout 300 python bench.py --config c4 --steps 20 --warmup 3 > gpurun_out/r6b_c4.json 2> gpurun_out/r6b_c4.err; echo "c4 rc=$?"; tail -c 1500 gpurun_out/r6b_c4.json
timeout 600 python bench.py --steps 20 --warmup 5 --extra-legs 0 --cpu-sample 0 > gpurun_out/r6b_bench_quick.json 2> gpurun_out/r6b_bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6b_bench_quick.json'))
print(d['value'], d['ms_per_step'], d.get('state_construction'), d['config'].get('verified'), d['roofline']['kernels_ms_per_step'])
PY
./tools/pool_trace.sh r6b 8 --no-poisson
./tools/pool_trace.sh r6b 1 --no-poisson
./tools/pool_trace.sh r6bp 8
