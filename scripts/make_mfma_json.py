#!/usr/bin/env python3
"""Matrix-pipe counters per launch from the rocprofv3 SQ summaries (scripts/rocprof_summary.py output of the sq2 and sq3 passes of
scripts/collect_profiles*.sh): profiles/rNN_mfma_counters.json, read by bench.py for `roofline.mfma`.
  insts        SQ_INSTS_MFMA                      matrix instructions per launch (v_mfma_f32_16x16x32_f16: 16 384 flop each)
  busy_cycles  SQ_VALU_MFMA_BUSY_CYCLES           summed over the 1024 SIMDs (cycles; 16 per v_mfma_f32_16x16x32_f16)
  busy_frac    busy_cycles / (1024 x GRBM_GUI_ACTIVE / 8)   share of the kernel's cycles the matrix pipes were busy (rocprofv3 sums
               GRBM_GUI_ACTIVE over the 8 XCDs: 1.07 ms x 2.0 GHz x 8 at 6144^2 / 31 x 31)
  valu_per_mfma (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA,  lds_per_mfma  SQ_INSTS_LDS / SQ_INSTS_MFMA
  issued_flops insts x 16384;  useful_flops = the algorithmic multiply-adds of the stage x 2 (2 K^2 3 per pixel and
               convolution; the fused kernel: one convolution + the gradient = 2 x that), i.e. WITHOUT the three-term split and the
               Toeplitz padding that the issued figure contains
    make_mfma_json.py sq2.txt sq3.txt --size N --psf K"""
import json
import re
import sys


def counters(path, names, kernels):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = next((v for k, v in kernels.items() if k in line), None) if "avg_us" not in line and len(line.split()) < 8 else None
        elif cur:
            for n in names:
                if re.match(r"\s+%s\s" % n, line):
                    out.setdefault(cur, {})[n] = float(re.search(r"mean ([0-9.e+]+)", line).group(1))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
    opt = {sys.argv[i][2:]: int(sys.argv[i + 1]) for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    size, psf = opt.get("size", 4096), opt.get("psf", 15)
    kernels = {"k_conv_mfma<%d, 0," % psf: "synth_residual", "k_conv_mfma<%d, 1," % psf: "backproject", "k_gradk_mfma<": "psf_gradient",
               "k_synth_gradk<%d," % psf: "synth_gradk", "k_conv_fft<0,": "synth_residual", "k_conv_fft<1,": "backproject", "k_gradk_fft<": "psf_gradient"}
    sq2 = counters(args[0], ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS"], kernels)
    sq3 = counters(args[1], ["GRBM_GUI_ACTIVE"], kernels)
    conv_flops = 2.0 * psf * psf * 3 * size * size
    useful = {"synth_residual": conv_flops, "backproject": conv_flops, "psf_gradient": conv_flops, "synth_gradk": 2 * conv_flops}
    out = {}
    for k, c in sq2.items():
        n = c["SQ_INSTS_MFMA"]
        if not n:      # (the transform-tile kernels issue no matrix instructions: fp32 VALU butterflies)
            out[k] = {"insts": 0, "valu_insts": int(c["SQ_INSTS_VALU"]), "lds_insts": int(c["SQ_INSTS_LDS"]), "useful_flops": int(useful[k])}
            continue
        gui = sq3.get(k, {}).get("GRBM_GUI_ACTIVE")
        out[k] = {"insts": int(n), "busy_cycles": int(c["SQ_VALU_MFMA_BUSY_CYCLES"]),
                  "busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8.0), 4) if gui else None,
                  "valu_per_mfma": round((c["SQ_INSTS_VALU"] - n) / n, 2), "lds_per_mfma": round(c["SQ_INSTS_LDS"] / n, 2),
                  "issued_flops": int(n * 16384), "useful_flops": int(useful[k]), "useful_over_issued": round(useful[k] / (n * 16384), 3)}
    import os
    print(json.dumps({"_comment": __doc__.strip(), "workload": {"size": size, "psf": psf}, "commit": os.environ.get("ICS_COMMIT", "unrecorded"), "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
