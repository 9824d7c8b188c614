"""cProfile of PageDistortionStep.run on the C4-shaped page: where the host time of a page goes (tottime, top 45)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from numpy.random import default_rng
from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input as _synthetic_page_input
from vkit_amd import _native as N
from vkit_amd.pipeline import text_detection as T
ctx = N.default_ctx()
step_input = _synthetic_page_input(seed=3, size=1024, n_lines=64)
assembler = T.page_assembler_step_factory.create()
distortion = T.page_distortion_step_factory.create()
page_out = assembler.run(step_input, default_rng(0))
dist_in = T.PageDistortionStepInput(page_out)
for s in range(8):
    distortion.run(dist_in, default_rng(s))
seeds = list(range(100, 148))
t = []
for s in seeds:
    t0 = time.perf_counter(); distortion.run(dist_in, default_rng(s)); t.append(time.perf_counter() - t0)
t.sort()
print('mean %.3f ms median %.3f ms max %.3f ms' % (sum(t) / len(t) * 1e3, t[len(t) // 2] * 1e3, t[-1] * 1e3))
pr = cProfile.Profile(); pr.enable()
for s in seeds:
    distortion.run(dist_in, default_rng(s))
pr.disable()
for key in ('tottime', 'cumulative'):
    buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
    print(buf.getvalue().replace(ROOT + '/', ''))
