"""Process placement next to the GPU (vkit_amd/shard.py: numa_cpus_of_pci, ranks_sharing_numa, bind_to_device_numa) on a made-up
sysfs tree: 8 GPUs on 2 NUMA nodes, like the node the scaling bench runs on.  The reference's pool leaves placement to the OS
(vkit/utility/pool.py:153-243)."""
import os

import pytest

from vkit_amd import shard


def _fake_sysfs(tmp_path, nodes, devices):
    for node, cpulist in nodes.items():
        d = tmp_path / 'devices' / 'system' / 'node' / f'node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cpulist + '\n')
    for bdf, node in devices.items():
        d = tmp_path / 'bus' / 'pci' / 'devices' / bdf
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(f'{node}\n')
    return str(tmp_path)


def test_cpulist_and_numa_lookup(tmp_path):
    sysfs = _fake_sysfs(tmp_path, {0: '0-3,64-67', 1: '4-7,68-71'}, {'0000:05:00.0': 0, '0000:c5:00.0': 1, '0000:e5:00.0': -1})
    assert shard.numa_cpus_of_pci('0000:05:00.0', sysfs) == {0, 1, 2, 3, 64, 65, 66, 67}
    assert shard.numa_cpus_of_pci('C5:00.0', sysfs) == {4, 5, 6, 7, 68, 69, 70, 71}      # short form, upper case
    assert shard.numa_cpus_of_pci('0000:e5:00.0', sysfs) is None                          # the kernel does not say
    assert shard.numa_cpus_of_pci('0000:ff:00.0', sysfs) is None                          # no such device


def test_eight_gpus_on_two_nodes_share_the_cores_of_their_node(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 8:
        pytest.skip('needs 8 CPUs')
    half = len(allowed) // 2
    lists = {0: ','.join(str(c) for c in allowed[:half]), 1: ','.join(str(c) for c in allowed[half:])}
    bus = [f'0000:{k:02x}:00.0' for k in range(8)]
    sysfs = _fake_sysfs(tmp_path, lists, {b: (0 if k < 4 else 1) for k, b in enumerate(bus)})
    assert [shard.ranks_sharing_numa(bus, k, sysfs) for k in range(8)] == [(0, 4), (1, 4), (2, 4), (3, 4)] * 2
    before = os.sched_getaffinity(0)
    try:
        seen = []
        for k in range(8):
            os.sched_setaffinity(0, before)
            pos, sharing = shard.ranks_sharing_numa(bus, k, sysfs)
            got = shard.bind_to_device_numa(bus[k], pos, sharing, sysfs)
            assert got == os.sched_getaffinity(0)
            node_cpus = set(allowed[:half] if k < 4 else allowed[half:])
            assert got <= node_cpus and len(got) == max(1, len(node_cpus) // 4)
            seen.append(frozenset(got))
        assert len(set(seen)) == 8 and not any(a & b for i, a in enumerate(seen) for b in seen[i + 1:])     # disjoint shares
        # unknown topology / switched off: left alone
        os.sched_setaffinity(0, before)
        assert shard.bind_to_device_numa('0000:ff:00.0', 0, 1, sysfs) is None and os.sched_getaffinity(0) == before
        os.environ['VKX_NO_AFFINITY'] = '1'
        assert shard.bind_to_device_numa(bus[0], 0, 4, sysfs) is None and os.sched_getaffinity(0) == before
    finally:
        os.environ.pop('VKX_NO_AFFINITY', None)
        os.sched_setaffinity(0, before)
