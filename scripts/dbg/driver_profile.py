"""cProfile of one device-resident deblur_module run (after a warm-up run): where the HOST spends its time between the solver calls.
   python scripts/dbg/driver_profile.py [size] [blur_width] [iterations]"""
import contextlib, cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-cases-studies_amd"))
import deconvolve as dv
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 15
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(0)
coarse = rng.random((size // 8 + 2, size // 8 + 2, 3))
pic = (np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:size, :size] * 200 + 20).astype(np.uint8)
kw = dict(mask=[size // 2, size // 2], mask_size=255, display=False, iterations=iters, save=False)
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        return dv.deblur_module(pic, "t", ".", bw, device_resident=True, **kw)
run(); run()
ts = []
for _ in range(3):
    t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
print("resident runs: " + ", ".join("%.1f ms" % (1e3 * x) for x in ts))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("tottime").print_stats(22)
