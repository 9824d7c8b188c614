"""k_composite_rgb (uint8 RGB pages whose rows are whole 4-pixel groups) at every run length of tile slots per workgroup: the
batched launch of several pages against the oracle's sequential fill_np_array, layer kinds mixed -- float alpha planes, byte masks,
value planes, scalar alphas, the keep-max / keep-min copies -- boxes that cut 4-pixel groups, and more (tile, layer) pairs in a run than
one pass stages.  VKX_RGB_RUN is read once per process: one child process per run length."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, json, sys
import numpy as np
from numpy.random import default_rng
sys.path.insert(0, %(root)r)
import oracle as O
from vkit_amd import _native as N

ctx, lib = N.default_ctx(), N.lib()
rng = default_rng(%(seed)d)
h, w, cn = 200, 256, 3
n_pages = 5
pages = [rng.integers(0, 256, (h, w, cn), dtype=np.uint8) for _ in range(n_pages)]
counts = [70, 1, 0, 45, 12]          # 70 layers, many on the same tiles: more pairs in a run than one pass stages
layers = (N.VkxLayer * sum(counts))()
keep, specs, begin = [], [], [0]
k = 0
for p, cnt in enumerate(counts):
    for j in range(cnt):
        if p == 0 and j %% 2 == 0:      # stacked on the first tiles
            bh, bw = int(rng.integers(8, 40)), int(rng.integers(60, 200))
            up, left = int(rng.integers(0, 24)), int(rng.integers(0, 50))
        else:
            bh, bw = int(rng.integers(1, h)), int(rng.integers(1, w))
            up, left = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
        if p == 3 and j == 0:           # an opaque first layer over the whole page: no destination read
            up, left, bh, bw = 0, 0, h, w
        kind = int(rng.integers(6)) if not (p == 3 and j == 0) else 5
        L = layers[k]
        L.up, L.left, L.height, L.width = up, left, bh, bw
        color = tuple(int(v) for v in rng.integers(0, 256, 3))
        for c in range(3):
            L.value_const[c] = color[c]
        alpha = mask = value = None
        a, mode = 1.0, 0
        if kind == 0:
            alpha = (rng.random((bh, bw), dtype=np.float32) * (rng.random((bh, bw)) < 0.5)).astype(np.float32)
            d = ctx.malloc(alpha.nbytes); ctx.upload(d, alpha); keep.append(d)
            L.alpha, L.alpha_stride_el = d, bw
        elif kind == 1:
            mask = (rng.random((bh, bw)) < 0.4).astype(np.uint8)
            d = ctx.malloc(mask.nbytes); ctx.upload(d, mask); keep.append(d)
            L.mask, L.mask_stride = d, bw
            a = float(rng.choice([1.0, 0.35]))
        elif kind == 2:
            value = rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
            d = ctx.malloc(value.nbytes); ctx.upload(d, value); keep.append(d)
            L.value, L.value_stride = d, bw * 3
            a = float(rng.choice([1.0, 0.6]))
        elif kind == 3:
            mode = int(rng.choice([1, 2]))          # keep max / keep min: scalar alpha 1.0
            if rng.random() < 0.5:
                value = rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
                d = ctx.malloc(value.nbytes); ctx.upload(d, value); keep.append(d)
                L.value, L.value_stride = d, bw * 3
        elif kind == 4:
            a = float(rng.choice([0.6, 0.0, 0.25]))
        L.alpha_scalar, L.mode = a, mode
        specs.append((p, (up, left, bh, bw), value if value is not None else color, alpha, mask, a, mode))
        k += 1
    begin.append(k)
d_pages = [ctx.malloc(pg.nbytes) for pg in pages]
for dp, pg in zip(d_pages, pages):
    ctx.upload(dp, pg)
ptrs = (ctypes.c_void_p * n_pages)(*d_pages)
b = np.asarray(begin, np.int32)
N.check(lib.vkx_fill_u8_batch_dev(ctx.handle, ptrs, n_pages, h, w, cn, w * cn, layers, b.ctypes.data))
bad = []
for p in range(n_pages):
    got = np.empty_like(pages[p])
    ctx.download(d_pages[p], got); ctx.sync()
    want = pages[p].copy()
    for (pp, box, value, alpha, mask, a, mode) in specs:
        if pp == p:
            O.fill(want, box, value, mask=mask, alpha=alpha if alpha is not None else a, mode=mode)
    if not np.array_equal(got, want):
        bad.append((p, int((got != want).sum())))
print(json.dumps({'bad': bad, 'layers': k}))
'''


@pytest.mark.gpu
@pytest.mark.parametrize('run', [1, 2, 3, 8])
def test_composite_rgb_run_lengths(run):
    env = dict(os.environ, VKX_RGB_RUN=str(run))
    for seed in (3, 4):
        proc = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'seed': seed}], capture_output=True, text=True, env=env, timeout=300)
        assert proc.returncode == 0, proc.stderr[-2000:]
        res = json.loads(proc.stdout.strip().splitlines()[-1])
        assert res['bad'] == [], (run, seed, res)
