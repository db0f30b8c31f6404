#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time stats and PMC counter means."""
import sqlite3, sys, collections
def short(n):
    for a,b in (("_ZN12_GLOBAL__N_1",""),("IcsConvArgs",""),("Ev11",""),):
        n=n.replace(a,b)
    return n[:70]
def main(path):
    con=sqlite3.connect(path); cur=con.cursor()
    cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows=cur.execute("select name, start, end from kernels").fetchall()
    agg=collections.defaultdict(list)
    for n,s,e in rows: agg[n].append((e-s)/1e3)
    print("%-72s %6s %10s %10s %10s"%("kernel","calls","avg_us","min_us","total_us"))
    for n,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        print("%-72s %6d %10.1f %10.1f %10.1f"%(short(n),len(v),sum(v)/len(v),min(v),sum(v)))
    # idle time of the device between consecutive kernels, by the kernel that FOLLOWS the gap (launch gaps, host round trips)
    tl=sorted((s,e,n) for n,s,e in rows)
    gaps=collections.defaultdict(list); busy_end=None
    for s_,e_,n in tl:
        if busy_end is not None: gaps[n].append(max(0,(s_-busy_end))/1e3)
        busy_end=e_ if busy_end is None else max(busy_end,e_)
    if tl:
        span=(max(e for _,e,_ in tl)-tl[0][0])/1e3; busy=sum(sum(v) for v in agg.values())
        print("timeline %.1f us, kernels %.1f us, idle %.1f us (%.1f %%)"%(span,busy,span-busy,100*(span-busy)/span))
        print("%-72s %6s %10s %10s"%("idle before kernel","gaps","avg_us","total_us"))
        for n,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1]))[:14]:
            print("%-72s %6d %10.2f %10.1f"%(short(n),len(v),sum(v)/len(v),sum(v)))
    import os
    if os.environ.get("ICS_GAPS"):   # every idle gap above a threshold (us), with its neighbours
        thr=float(os.environ["ICS_GAPS"]); prev=None; pn=""; k=0
        for s_,e_,n in tl:
            if prev is not None and (s_-prev)/1e3>thr and k<60: print("   gap %8.2f us after %-40s before %s"%((s_-prev)/1e3,short(pn)[-40:],short(n)[:50])); k+=1
            prev=e_ if prev is None else max(prev,e_); pn=n
    if os.environ.get("ICS_SEQ"):   # a window of the timeline: gap before, duration
        mid=len(tl)//2; prev=None
        for s_,e_,n in tl[mid-1:mid+int(os.environ["ICS_SEQ"])]:
            if prev is not None: print("   gap %8.2f us   run %8.2f us   %s"%((s_-prev)/1e3,(e_-s_)/1e3,short(n)[:60]))
            prev=e_
    try:
        pc=[r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        rows=cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    except Exception as ex:
        print("no counters:",ex); return
    c=collections.defaultdict(lambda: collections.defaultdict(list))
    for k,cn,v in rows: c[k][cn].append(v)
    for k,d in c.items():
        print(short(k))
        for cn,v in sorted(d.items()): print("    %-28s mean %.4g  (n=%d)"%(cn,sum(v)/len(v),len(v)))
main(sys.argv[1])
