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
