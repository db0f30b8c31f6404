"""A/B of two builds of the library on one GPU box, alternating processes:
   python scripts/dbg/ab_libs.py LIB_A LIB_B SIZE PSF BLIND [STEPS] [REPS]   (ICS_HIP_LIB selects the build; scripts/dbg/time_config.py does the timing)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la, lb, size, psf, blind = sys.argv[1:6]
steps = sys.argv[6] if len(sys.argv) > 6 else "1000"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
for rep in range(reps):
    for lib in (la, lb):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dbg", "time_config.py"), size, psf, blind, steps],
                             env=dict(os.environ, ICS_HIP_LIB=os.path.join(ROOT, lib)), capture_output=True, text=True)
        print(os.path.basename(lib), (out.stdout.strip().splitlines() or ["FAILED " + out.stderr[-400:]])[-1], flush=True)
