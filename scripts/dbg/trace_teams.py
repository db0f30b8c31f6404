"""End times of the first vs second half of the grid, and per-tile durations over time: python scripts/dbg/trace_teams.py trace.bin [tile_mark]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, collections
from trace_conv_mfma import load
W = load(sys.argv[1]); TM = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = min(v['ev'][0][0] for v in W.values())
nwg = max(W) // 4 + 1
for name, sel in (("first half of the grid ", lambda b: b < nwg // 2), ("second half of the grid", lambda b: b >= nwg // 2)):
    e = np.array([(v['ev'][-1][0] - t0) / 100 for w, v in W.items() if sel(w // 4)])
    print("%s: %4d waves, ends p10 %.0f p50 %.0f p90 %.0f" % (name, len(e), *np.percentile(e, [10, 50, 90])))
# CU mates: pairs of workgroups on the same CU
cu = collections.defaultdict(set)
for w, v in W.items(): cu[(v['xcc'], v['se'], v['sh'], v['cu'])].add(w // 4)
pairs = [sorted(s) for s in cu.values() if len(s) == 2]
d = np.array([b - a for a, b in pairs])
print("CUs with two workgroups: %d; blockIdx difference of the mates: %s" % (len(pairs), collections.Counter(d.tolist()).most_common(4)))
endwg = {}
for w, v in W.items(): endwg[w // 4] = max(endwg.get(w // 4, 0), (v['ev'][-1][0] - t0) / 100)
first = np.array([endwg[a] for a, b in pairs]); second = np.array([endwg[b] for a, b in pairs])
print("the mate with the LOWER blockIdx ends at p50 %.0f, the other at p50 %.0f; lower one first on %d of %d CUs" % (np.median(first), np.median(second), int((first < second).sum()), len(pairs)))
# tile durations of one wave per workgroup over the walk
for name, sel in (("lower ", [a for a, b in pairs]), ("higher", [b for a, b in pairs])):
    T = []
    for b in sel:
        ev = W[b * 4]['ev']; ts = [t for t, m in ev if m == TM]
        T.append(np.diff([ev[0][0]] + ts) / 100)
    L = min(len(x) for x in T); T = np.array([x[:L] for x in T])
    print("%s-blockIdx mate, us per tile along the walk: " % name + " ".join("%.0f" % x for x in T.mean(0)))
