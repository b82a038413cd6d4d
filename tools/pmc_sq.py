#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc passes of SQ counters per kernel (sums over dispatches) and derives the utilisation figures
DESIGN.md quotes.  Usage: pmc_sq.py [--json out.json] dir1 dir2 ...

Units (MI355X_MICROARCH.md, per-instruction constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_IDX_ACTIVE count cycles summed over SIMDs / CUs; SQ_BUSY_CYCLES is
summed over the 32 shader engines (wall cycles of the kernel = SQ_BUSY_CYCLES / 32).  Derived:
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (wall cycles x 1024 SIMDs)
  valu_util  = 4 x SQ_ACTIVE_INST_VALU  / (wall cycles x 1024 SIMDs)
  lds_util   = SQ_LDS_IDX_ACTIVE        / (wall cycles x 256 CUs)
  wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES  (waves parked at s_waitcnt / s_barrier)"""
import collections, csv, glob, json, re, sys
args = sys.argv[1:]
out_json = None
if args and args[0] == "--json":
    out_json, args = args[1], args[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in args:
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rln::", "").replace("rln::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add((d, r["Dispatch_Id"]))
names = sorted({c for v in agg.values() for c in v})
top = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]
summary = {}
for k, v in top:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    wall = v.get("SQ_BUSY_CYCLES", 0) / 32.0
    der = {}
    if wall > 0:
        der = {"mfma_util": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (wall * 1024),
               "valu_util": 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / (wall * 1024),
               "lds_util": v.get("SQ_LDS_IDX_ACTIVE", 0) / (wall * 256),
               "lds_conflict_share": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1),
               "wait_share": v.get("SQ_WAIT_ANY", 0) / wc,
               "issue_stall_share": v.get("SQ_WAIT_INST_ANY", 0) / wc}
    summary[k] = {"counters": dict(v), "derived": {a: round(b, 4) for a, b in der.items()}}
    print(f"\n== {k[:70]}   SQ_WAVE_CYCLES {wc:.3e}   " + "  ".join(f"{a} {b:.3f}" for a, b in der.items()))
    for c in names:
        if c in v and c != "SQ_WAVE_CYCLES":
            print(f"   {c:30s} {v[c]:.4e}   {v[c] / wc:8.3f} per wave-cycle")
if out_json:
    json.dump(summary, open(out_json, "w"), indent=1, sort_keys=True)
