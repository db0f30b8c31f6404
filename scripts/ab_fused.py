"""A/B on the GPU box: blind inner iteration with the fused A11+A13 kernel vs the two-kernel path (same binary)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
size = sys.argv[1] if len(sys.argv) > 1 else "4096"
for env in ({"ICS_FUSED_GRADK": "1"}, {"ICS_FUSED_GRADK": "0"}):
    for rep in range(2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-mode", "--steps", "200", "--warmup", "20", "--size", size],
                             env=dict(os.environ, **env), capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(env, d["ms_per_step"], {k: v["ms"] for k, v in d["kernels_ms"].items()})
        except Exception:
            print(env, "FAILED", out.stderr[-800:])
