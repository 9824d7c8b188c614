"""``bench.py --config c4`` (BASELINE configs[3], page synthesis at N GPUs) on one GPU: the resident batch is checked against the oracle
inside the run, the reference-API leg runs the three steps, and the line carries the evidence an N > 1 reader needs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c4_mode_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'c4', '--batch', '6', '--steps', '3', '--warmup', '1',
                          '--api-seconds', '1.0', '--api-workers', '2'], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert rec['unit'] == 'pages/s' and rec['n_gpus'] == 1 and rec['steps'] == 3
    assert rec['config']['verified_against_oracle'] == 2            # first and last page of the rank: composite and chain
    assert abs(rec['value'] - 6 * 3 / (rec['ms_per_step'] * 3 / 1e3)) < 1e-6 * rec['value']
    assert 'k_composite_rgb' in rec['kernels_ms_per_step'] and 'k_chain_fused' in rec['kernels_ms_per_step']
    api = rec['reference_api']
    assert api['pages'] >= 10 and api['workers'] == 2 and api['workers_per_gpu'] == 2 and api['pages_per_s'] > 20
    ev = rec['config']['distributed']
    assert ev['world_size'] == 1 and len(ev['ranks']) == 1 and ev['ranks'][0]['pages'] == 6
