"""``python bench.py --gpus N`` from a plain shell starts its own ranks (the reference's pool forks its workers itself,
vkit/utility/pool.py:153-243): rendezvous on 127.0.0.1, barriers, MAX reduction and the rank-0 JSON line, here over gloo without
a GPU (``--dry-run`` leaves out everything but the protocol)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, capture_output=True, text=True, timeout=300,
                          env=env)


def test_plain_invocation_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = _run(['--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1', '--batch', '5'], env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout                     # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec['dry_run'] and rec['n_gpus'] == 2 and rec['steps'] == 3
    assert rec['units'] == 2 * 5 * 3                       # SUM over the ranks: every rank ran its own 5 images 3 times
    assert rec['first_image_of_last_rank'] == 5            # weak scaling: rank r owns images [r B, (r + 1) B)
    assert rec['ms_per_step'] >= 20.0                      # MAX over the ranks: rank 1 sleeps 20 ms per step, rank 0 10 ms


def test_a_mismatched_world_is_refused():
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    out = _run(['--gpus', '2', '--dry-run'], env)
    assert out.returncode != 0 and 'WORLD_SIZE=3' in (out.stdout + out.stderr)


def test_single_rank_dry_run_needs_no_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = _run(['--dry-run', '--steps', '2', '--warmup', '0', '--batch', '4'], env)
    assert out.returncode == 0, out.stdout + out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec['n_gpus'] == 1 and rec['units'] == 8


def test_eight_ranks_rendezvous_and_reduce():
    """The launch the scaling bench makes on the 8-GPU node (``python bench.py --gpus 8``), protocol only: eight ranks over gloo on the
    loopback address, one JSON line from rank 0, units summed and times maximised over all eight; the same for the PCIe-inclusive mode's
    command line (``--mode dropin``; a dry run stops before any device work)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    for extra in ([], ['--mode', 'dropin'], ['--config', 'c4']):
        out = _run(['--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '1', '--batch', '3'] + extra, env)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, out.stdout
        rec = json.loads(lines[0])
        assert rec['n_gpus'] == 8 and rec['units'] == 8 * 3 * 2 and rec['first_image_of_last_rank'] == 21
        assert rec['ms_per_step'] >= 80.0                  # rank 7 sleeps 80 ms per step
        assert rec['mode'] == (extra[1] if extra and extra[0] == '--mode' else 'resident')
        assert rec['config'] == (extra[1] if extra and extra[0] == '--config' else 'c3')
        # the evidence every N > 1 line carries: the backend and world size the process group itself reports, one record per rank
        ev = rec['distributed']
        assert ev['backend'] == 'gloo' and ev['world_size'] == 8
        assert sorted(r['rank'] for r in ev['ranks']) == list(range(8)) and len({r['pid'] for r in ev['ranks']}) == 8
        assert [r['first_unit'] for r in ev['ranks']] == [3 * k for k in range(8)]
