"""The CPU legs of bench.py: the host facts it prints next to the CPU baseline and the process sweep (on a tiny workload)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_sweep_points_and_host_facts():
    import bench
    assert bench.sweep_points(1) == [1]
    assert bench.sweep_points(8) == [1, 2, 4, 8]
    assert bench.sweep_points(12) == [1, 2, 4, 8, 12]
    assert bench.sweep_points(256) == [1, 2, 4, 8, 16, 32, 64, 128, 256]
    facts = bench.host_cpu_facts()
    assert facts['os_cpu_count'] == os.cpu_count()
    assert facts['sched_getaffinity_at_start'] == len(os.sched_getaffinity(0)) or facts['sched_getaffinity_at_start'] >= 1
    assert 'cgroup_cpu_max' in facts and 'loadavg' in facts
    if facts.get('cgroup_cpu_max') and not facts['cgroup_cpu_max'].startswith('max'):
        assert facts['cgroup_quota_cpus'] > 0


def test_process_sweep_on_a_small_workload():
    """Two points (1 and 2 processes of one pool) at 256^2: the first point is the one-process figure, every point reports its own
    process count, the best point is the value, effective_cores = best / one process."""
    import bench
    out = bench.cpu_baseline_all_cores(256, 2, 1, single_thread_value=None)
    assert out is not None
    assert [e['processes'] for e in out['sweep']] == [1, 2]
    assert out['one_process_value'] == out['sweep'][0]['value'] > 0
    assert out['value'] == max(e['value'] for e in out['sweep'])
    assert abs(out['effective_cores'] - round(out['value'] / out['one_process_value'], 1)) < 1e-9
    assert out['knee_processes'] in (1, 2) and out['host']['os_cpu_count'] == os.cpu_count()
