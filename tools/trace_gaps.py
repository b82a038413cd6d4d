"""Idle time between consecutive kernels of a rocprofv3 kernel_trace.csv: per-step busy / idle split and the kernels that
precede the largest share of idle time.  usage: trace_gaps.py kernel_trace.csv steps [skip_fraction]"""
import csv, sys, collections, re
f, steps = sys.argv[1], int(sys.argv[2])
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                 re.sub(r'\(.*', '', r['Kernel_Name']).replace('void rln::', '').replace('rln::', '')[:40]))
rows.sort()
# steady state: the last 60 % of the trace (setup, warm-up and data generation come first)
n0 = int(len(rows) * 0.4)
rows = rows[n0:]
span = rows[-1][1] - rows[0][0]
busy = 0
gaps = collections.defaultdict(lambda: [0, 0])
cur_end = rows[0][0]
for i, (s, e, name) in enumerate(rows):
    if s > cur_end:
        g = s - cur_end
        prev = rows[i - 1][2] if i else '-'
        gaps[(prev, name)][0] += 1
        gaps[(prev, name)][1] += g
        busy += e - s
    else:
        busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
idle = span - busy
print(f"window {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {idle/1e6:.2f} ms ({100.0*idle/span:.1f} %), kernels {len(rows)}")
tot = sum(v[1] for v in gaps.values())
print(f"gaps: {sum(v[0] for v in gaps.values())} totalling {tot/1e6:.2f} ms; mean {tot/max(1,sum(v[0] for v in gaps.values()))/1e3:.2f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k[0]:40s} -> {k[1]:40s} n {v[0]:5d} mean {v[1]/v[0]/1e3:7.2f} us total {v[1]/1e6:6.3f} ms")
