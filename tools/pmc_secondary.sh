#!/bin/bash
# PMC tables of the secondary kernels: which counter caps each of them
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
./tools/pmc_any.sh r6_c5 k_tile_remap python tools/c5.py > gpurun_out/r6_pmc_c5_tile_remap.txt 2>&1
./tools/pmc_any.sh r6_k8192 k_ python tools/kernels.py 8192 > gpurun_out/r6_pmc_kernels_8192.txt 2>&1
./tools/pmc_any.sh r6_pz k_pz python tools/poisson_probe.py 1024 > gpurun_out/r6_pmc_poisson.txt 2>&1
rm -rf gpurun_out/pmc_r6_c5 gpurun_out/pmc_r6_k8192 gpurun_out/pmc_r6_pz
wc -l gpurun_out/r6_pmc_*.txt
timeout 120 python tools/kernels.py 8192 > gpurun_out/r6_kernels_8192.json 2>/dev/null
timeout 120 python tools/c5.py > gpurun_out/r6_c5.json 2>/dev/null
