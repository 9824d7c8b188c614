#!/usr/bin/env python3
"""Times the host-visible cost of building geometric states (generate_state <= 5 ms at 2048^2).
Usage: tools/state_timing.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from numpy.random import default_rng

from vkit_amd.mechanism import distortion as D
from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam, mls as P_mls

out = {}
for name, op, gen_cls, cfg_cls in (
        ('similarity_mls', D.similarity_mls, P_mls.SimilarityMlsConfigGenerator, P_mls.SimilarityMlsConfigGeneratorConfig),
        ('camera_cubic_curve', D.camera_cubic_curve, P_cam.CameraCubicCurveConfigGenerator, P_cam.CameraCubicCurveConfigGeneratorConfig)):
    for hw in (1024, 2048, 4096):
        cfg = gen_cls(cfg_cls(), 5)((hw, hw), default_rng(0))
        op.generate_state(cfg, (hw, hw))     # warm-up (context, scratch)
        ts = []
        for _ in range(20):
            t = time.perf_counter()
            st = op.generate_state(cfg, (hw, hw))
            ts.append(time.perf_counter() - t)
        ts.sort()
        out[f'{name}_{hw}'] = {'median_ms': round(ts[len(ts) // 2] * 1e3, 3), 'min_ms': round(ts[0] * 1e3, 3),
                               'vertices': int(st.src_image_grid.vertices.shape[0] * st.src_image_grid.vertices.shape[1])}
        print(name, hw, out[f'{name}_{hw}'], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(out, f, indent=1)
