"""cProfile of one deblur_module run with the frames on the host between the solver calls (device_resident=False)."""
import contextlib, cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import deconvolve as dv
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
coarse = rng.random((size // 8 + 2, size // 8 + 2, 3))
pic = (np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:size, :size] * 200 + 20).astype(np.uint8)
kw = dict(mask=[size // 2, size // 2], mask_size=255, display=False, iterations=20, save=False, device_resident=False)
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        return dv.deblur_module(pic, "t", ".", 15, **kw)
run()
t = time.perf_counter(); run(); print("host-frame run: %.3f s" % (time.perf_counter() - t))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(16)
