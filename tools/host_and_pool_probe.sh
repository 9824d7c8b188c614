#!/bin/bash
# Round 6, first GPU job: cv2 attempt, host CPU facts + process sweep, page profile, pool scaling (>= 5 s per point, processes and threads)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
./tools/cv2_attempt.sh
{ nproc; lscpu | head -25; cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpuset.cpus.effective 2>&1; cat /sys/fs/cgroup/cpu.stat 2>&1; free -g | head -3; } > gpurun_out/r6_host_facts.txt 2>&1
timeout 400 python tools/cpu_sweep.py 2048 12 > gpurun_out/r6_cpu_sweep.json 2> gpurun_out/r6_cpu_sweep.err; echo "sweep rc=$?"
timeout 300 python tools/probes/page_profile.py 96 > gpurun_out/r6a_page_profile.txt 2>&1; echo "profile rc=$?"
timeout 900 python tools/pool_scale.py --workers 1,2,4,8,16 --seconds 5 --kind processes,threads --modes pipeline > gpurun_out/r6a_pool_scale.json 2> gpurun_out/r6a_pool_scale.err; echo "pool rc=$?"
timeout 400 python tools/pool_scale.py --workers 1,4,8 --seconds 5 --kind processes --modes pipeline --no-poisson > gpurun_out/r6a_pool_scale_nopoisson.json 2> gpurun_out/r6a_pool_scale_nopoisson.err; echo "pool-np rc=$?"
tail -3 gpurun_out/r6a_pool_scale.err
