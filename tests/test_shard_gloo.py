"""N>1 path on CPU: two processes over gloo run the sharding / timing protocol that bench.py uses with RCCL."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from vkit_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spans_partition_the_work():
    for total in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard.strong_span(total, r, world) for r in range(world)]
            covered = [i for first, count in spans for i in range(first, first + count)]
            assert covered == list(range(total))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert [shard.weak_span(256, r) for r in range(3)] == [(0, 256), (256, 256), (512, 256)]
    assert [shard.device_for(r, 8) for r in (0, 7, 8, 11)] == [0, 7, 0, 3]
    with pytest.raises(ValueError):
        shard.strong_span(10, 2, 2)
    with pytest.raises(RuntimeError):
        shard.device_for(0, 0)


_WORKER = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, os.environ['VKX_ROOT'])
    from vkit_amd import shard

    group = shard.Group(backend='gloo')
    first, count = shard.weak_span(5, group.rank)
    done = []

    def step():
        # stand-in for one pass over this rank's images; rank 1 is the slow one
        done.append(list(range(first, first + count)))
        time.sleep(0.05 * (1 + 3 * group.rank))

    elapsed = shard.timed_steps(group, step, steps=2, warmup=1, device_sync=lambda: None)
    units = group.sum_int(count * 2)
    own = time.perf_counter()
    out = {'rank': group.rank, 'world': group.world, 'first': first, 'count': count, 'passes': len(done),
           'elapsed': elapsed, 'units': units}
    with open(os.path.join(os.environ['VKX_OUT'], f'rank{group.rank}.json'), 'w') as fout:
        json.dump(out, fout)
    group.close()
''')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_ranks_over_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), VKX_ROOT=ROOT, VKX_OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    outs = [json.loads((tmp_path / f'rank{r}.json').read_text()) for r in range(2)]
    assert [o['first'] for o in outs] == [0, 5] and all(o['count'] == 5 for o in outs)
    assert all(o['passes'] == 3 for o in outs)                 # 1 warmup + 2 timed
    assert all(o['units'] == 20 for o in outs)                 # whole-job units: both ranks, both steps
    # MAX over ranks: both report the slow rank's time (2 x 0.2 s), not their own
    assert outs[0]['elapsed'] == outs[1]['elapsed']
    assert 0.38 <= outs[0]['elapsed'] < 5.0
