import cProfile, pstats, io, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from numpy.random import default_rng
from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input as _synthetic_page_input
from vkit_amd.pipeline import text_detection as T
from vkit_amd import _native as N
step_input = _synthetic_page_input(seed=3, size=1024, n_lines=64)
assembler = T.page_assembler_step_factory.create()
dconf = {'random_distortion_factory_config': {'disabled_policy_names': ['poisson_noise']}}
distortion = T.page_distortion_step_factory.create(dconf)
resizing = T.page_resizing_step_factory.create()
def page(seed):
    rng = default_rng(seed)
    a = assembler.run(step_input, rng)
    d = distortion.run(T.PageDistortionStepInput(a), rng)
    r = resizing.run(T.PageResizingStepInput(d), rng)
    return int(r.page_image.mat[0,0,0])
for s in range(5): page(s)
pr = cProfile.Profile(); pr.enable()
for s in range(40): page(100+s)
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(28); print(out.getvalue()[:6000])
