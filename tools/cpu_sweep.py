#!/usr/bin/env python3
"""The CPU legs of bench.py alone (no GPU work): the 1-thread oracle sample and the process sweep with the host facts.
Usage: tools/cpu_sweep.py [size] [sample_images] [max_procs]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

if __name__ == '__main__':
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else len(bench.START_AFFINITY or [0])
    one = bench.cpu_baseline(size, sample)
    one['host'] = bench.host_cpu_facts()
    one['all_cores'] = bench.cpu_baseline_all_cores(size, procs, 2, one['value'])
    print(json.dumps(one))
