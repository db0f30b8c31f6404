#!/usr/bin/env python3
"""HBM bytes per launch from the rocprofv3 PMC summaries (scripts/rocprof_summary.py output of a --pmc FETCH_SIZE run and of a
--pmc WRITE_SIZE run of the same command): profiles/rNN_hbm_traffic.json, read by bench.py for `roofline.traffic`.
Counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of a wide coalesced read -> x2
(calibrated in round 1 on a 211.5 MB device copy: FETCH_SIZE 103 MB; WRITE_SIZE 206.5 MB -> x1)."""
import json
import re
import sys

def kernel_names(psf):
    return {"k_conv_mfma<%d, 0," % psf: "synth_residual", "k_conv_mfma<%d, 1," % psf: "backproject", "k_update_rows<0>": "update",
            "k_gradk_mfma<": "psf_gradient", "k_synth_gradk<%d," % psf: "synth_gradk",
            # the transform-tile pipeline (ics_conv_fft.hip, ics_planar.hip): its own key, `kernels_fft`
            "k_conv_fft<0,": "synth_residual", "k_conv_fft<1,": "backproject", "k_update_planar": "update", "k_gradk_fft<": "psf_gradient",
            # round 6: A1 + A3 as one unit per tile pair, A11 + A13 as one unit per tile pair
            "k_conv_fft<2,": "synth_backproject", "k_synth_gradk_fft<": "synth_gradk"}


FFT_NAMES = ("k_conv_fft<", "k_update_planar", "k_gradk_fft<", "k_synth_gradk_fft<")


KERNELS = {}
ALGO = {"synth_residual": 36, "backproject": 48, "update": 60, "psf_gradient": 24, "synth_gradk": 60, "synth_backproject": 84}   # bytes per pixel


def counters(path, name):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            if any(n in line for n in FFT_NAMES):
                counters.fft = True
            cur = next((v for k, v in KERNELS.items() if k in line), None) if "avg_us" not in line and len(line.split()) < 8 else None
        elif cur and name in line:
            m = re.search(r"mean ([0-9.e+]+)", line)
            out[cur] = float(m.group(1))
    return out


def main():
    """make_traffic_json.py fetch.txt write.txt [fetch_two_kernel.txt] [--size N] [--psf K]   (the workload the passes were run on)"""
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {sys.argv[i][2:]: int(sys.argv[i + 1]) for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    args = [a for a in args if not a.isdigit()]
    size, psf = opt.get("size", 4096), opt.get("psf", 15)
    KERNELS.update(kernel_names(psf))
    fetch, write = counters(args[0], "FETCH_SIZE"), counters(args[1], "WRITE_SIZE")
    px = size * size
    kern = {}
    for k in fetch:
        w = write.get(k, 0.0)
        kern[k] = {"FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": w, "hbm_bytes": int((2 * fetch[k] + w) * 1024), "algorithmic_bytes": ALGO[k] * px}
        kern[k]["ratio"] = round(kern[k]["hbm_bytes"] / kern[k]["algorithmic_bytes"], 3)
    import os
    key = "kernels_fft" if getattr(counters, "fft", False) else "kernels_matrix"
    out = {"_comment": __doc__.strip(), "workload": {"size": size, "psf": psf}, "commit": os.environ.get("ICS_COMMIT", "unrecorded"), key: kern}
    if len(args) > 2:   # the two-kernel gradient path (ICS_FUSED_GRADK=0), fetch side only
        f2 = counters(args[2], "FETCH_SIZE")
        out["two_kernel_gradient_path_fetch_KiB"] = {k: f2[k] for k in ("synth_residual", "psf_gradient") if k in f2}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
