"""Aggregates a rocprofv3 kernel_trace.csv by (kernel, grid) -> per-step time; grid identifies the resolution level."""
import csv, sys, collections, re
f, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void rln::', '').replace('rln::', '')
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[(name, g)][0] += 1
    agg[(name, g)][1] += d
tot = sum(v[1] for v in agg.values()) / steps
print(f"total kernel time {tot/1000:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0][:52]:52s} blocks {str(k[1]):18s} calls/step {v[0]/steps:6.1f} avg {v[1]/v[0]:8.1f} us  per-step {v[1]/steps/1000:6.2f} ms")
