#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc passes of SQ counters per kernel (sum over dispatches).  Usage: pmc_sq.py dir1 dir2 ..."""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rln::", "").replace("rln::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
top = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
for k, v in top:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"\n== {k[:70]}   SQ_WAVE_CYCLES {wc:.3e}")
    for c in names:
        if c in v and c != "SQ_WAVE_CYCLES":
            print(f"   {c:30s} {v[c]:.4e}   {v[c] / wc:8.3f} per wave-cycle")
