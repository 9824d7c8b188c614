#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6j_driver_line.json 2> gpurun_out/r6j_driver_line.err; echo "rc=$? wall $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d=json.load(open("gpurun_out/r6j_driver_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["verified"])
print(json.dumps(d["other_configs"]["C4_reference_api_worker_pool"])[:1200])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["all_cores"]["effective_cores"])
PY
