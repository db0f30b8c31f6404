"""Is a workgroup's speed a property of the CU it runs on?  End times per (xcc, se, cu) of two traced launches: python scripts/dbg/trace_cus.py a.bin b.bin"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, collections
from trace_conv_mfma import load
def per_cu(path):
    W = load(path); t0 = min(v['ev'][0][0] for v in W.values())
    d = collections.defaultdict(list)
    for v in W.values(): d[(v['xcc'], v['se'], v['sh'], v['cu'])].append((v['ev'][-1][0] - t0) / 100)
    return {k: max(x) for k, x in d.items()}, {k: len(x) for k, x in d.items()}
a, na = per_cu(sys.argv[1]); b, nb = per_cu(sys.argv[2])
keys = sorted(set(a) & set(b))
x = np.array([a[k] for k in keys]); y = np.array([b[k] for k in keys])
print("%d CUs in both launches (%d / %d); waves per CU: %s" % (len(keys), len(a), len(b), sorted(set(na.values()))))
print("end time per CU: launch A p10/p50/p90 %.0f %.0f %.0f, launch B %.0f %.0f %.0f; correlation %.3f" % (*np.percentile(x, [10, 50, 90]), *np.percentile(y, [10, 50, 90]), np.corrcoef(x, y)[0, 1]))
# by cu index within the shader array, by se
for name, idx in (("cu", 3), ("se", 1), ("sh", 2), ("xcc", 0)):
    g = collections.defaultdict(list)
    for k in keys: g[k[idx]].append((a[k] + b[k]) / 2)
    print("  by %-3s: " % name + "  ".join("%s:%.0f(%d)" % (i, np.mean(v), len(v)) for i, v in sorted(g.items())))
