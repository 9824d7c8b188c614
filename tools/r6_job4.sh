#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 200 python tools/probes/page_transfers.py 40 > gpurun_out/r6_page_transfers.txt 2>&1; echo rc=$?
./tools/probes/copy_kinds.sh
cat gpurun_out/r6_page_transfers.txt
