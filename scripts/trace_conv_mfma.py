"""Decode the phase timeline written by tools/bench_conv_mfma.hip -DICS_MFMA_TRACE (one record per wave)."""
import collections
import struct
import sys

import numpy as np

NAMES = {15: 'start', 8: 'weights', 0: 'convert', 7: 'prefetch', 1: 'mfma', 3: 'epi-issue', 5: 'epilogue', 6: 'loop-exit', 9: 'reductions'}


def load(path):
    d = open(path, 'rb').read()
    off = 0
    W = {}
    while off < len(d):
        w, n = struct.unpack_from('<QQ', d, off)
        off += 16
        t = np.frombuffer(d, dtype='<u8', count=n, offset=off)
        off += 8 * n
        hw = int(t[0])
        hwid = hw & 0xffffffff
        W[w] = dict(xcc=(hw >> 32) & 0xf, cu=(hwid >> 8) & 0xf, sh=(hwid >> 12) & 1, se=(hwid >> 13) & 7, simd=(hwid >> 4) & 3,
                    ev=[(int(x) >> 8, int(x) & 0xff) for x in t[1:]])
    return W


def main(path):
    W = load(path)
    t0 = min(v['ev'][0][0] for v in W.values())
    t1 = max(v['ev'][-1][0] for v in W.values())
    print('%d waves, span %.1f us' % (len(W), (t1 - t0) / 100.0))
    dur = collections.defaultdict(list)
    # occupancy of phases over time (10 ns bins -> 1 us bins)
    nb = (t1 - t0) // 100 + 2
    occ = {m: np.zeros(nb) for m in NAMES}
    for v in W.values():
        prev = v['ev'][0][0]
        for (t, m) in v['ev'][1:]:
            dur[m].append((t - prev) / 100.0)
            a, b = (prev - t0) / 100.0, (t - t0) / 100.0
            ia, ib = int(a), int(b)
            for i in range(ia, ib + 1):
                occ[m][i] += (min(b, i + 1) - max(a, i))
            prev = t
    for m in (8, 0, 7, 1, 3, 5, 6, 9):
        if not dur[m]: continue
        x = np.array(dur[m])
        print('%-10s n=%5d mean %.2f us  p10 %.2f p50 %.2f p90 %.2f' % (NAMES[m], len(x), x.mean(), *np.percentile(x, [10, 50, 90])))
    print('waves per phase over time (1-us bins): t, convert, prefetch, mfma, epi-issue, epilogue')
    for i in range(0, nb, 4):
        print('%4d ' % i + ' '.join('%6.0f' % occ[m][i] for m in (0, 7, 1, 3, 5)))


if __name__ == '__main__':
    main(sys.argv[1])
