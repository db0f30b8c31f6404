"""Does workgroup b run on XCD b % 8?  python scripts/dbg/trace_xcd.py trace.bin"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_conv_mfma import load
W = load(sys.argv[1])
c = collections.Counter(((w // 4) % 8 == v['xcc']) for w, v in W.items())
m = collections.Counter(((w // 4) % 8, v['xcc']) for w, v in W.items())
print("waves whose workgroup index mod 8 equals their XCC_ID: %d of %d" % (c[True], len(W)))
if c[False]: print("  (block mod 8, xcc) pairs:", sorted(m.items())[:16])
