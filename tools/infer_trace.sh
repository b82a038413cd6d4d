#!/bin/bash
# Kernel trace of the 480x640 eval forward (N = 1 and 16): is the latency kernel time or launch gaps?
R=/root/repo
OUT=$R/gpurun_out/infer_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/infer_bench.py > $OUT/bench.txt 2> $OUT/bench.log
python3 - <<PY > $OUT/summary.txt
import csv, glob, re, collections
f = glob.glob("$OUT/stats/*/*kernel_trace.csv")[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
# forwards are separated by nothing in particular: report per kernel class totals and the busy/idle split of the whole trace
busy = sum(e - s for s, e, _ in rows)
print("kernels", len(rows), "busy ms", busy / 1e6, "span ms", (rows[-1][1] - rows[0][0]) / 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = re.sub(r'\(.*', '', n).replace('void rln::', '').replace('rln::', '')[:50]
    agg[k][0] += 1; agg[k][1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:50s} n {v[0]:6d} total {v[1]/1e6:8.2f} ms avg {v[1]/v[0]/1e3:8.1f} us")
PY
rm -rf $OUT/stats
