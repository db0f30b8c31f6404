"""Per-XCD finish times of one traced launch (tools/bench_conv_mfma.hip -DICS_MFMA_TRACE): python scripts/dbg/trace_tail.py trace_mode0.bin"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trace_conv_mfma import load
W = load(sys.argv[1])
t0 = min(v['ev'][0][0] for v in W.values())
end = np.array([(v['ev'][-1][0] - t0) / 100 for v in W.values()]); xcc = np.array([v['xcc'] for v in W.values()])
start = np.array([(v['ev'][0][0] - t0) / 100 for v in W.values()])
TM = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ntile = np.array([sum(1 for _, m in v["ev"] if m == TM) for v in W.values()])
print("waves %d, span %.1f us; starts p50 %.1f max %.1f; ends p1 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (len(W), end.max(), np.median(start), start.max(), *np.percentile(end, [1, 10, 50, 90]), end.max()))
for x in sorted(set(xcc)):
    e = end[xcc == x]; n = ntile[xcc == x]
    print("  xcd %d: %4d waves, tiles per wave %4.1f (min %d max %d), ends min %.1f p50 %.1f max %.1f" % (x, len(e), n.mean(), n.min(), n.max(), e.min(), np.median(e), e.max()))
busy = sum(end - start) / (len(W) * end.max())
print("mean wave lifetime / span = %.3f (1 - that = tail + stagger)" % busy)
