"""Lists the per-dispatch durations of a kernel (name substring) from a rocprofv3 kernel_trace.csv for one step
(dispatch order = layer order), to see which layers/levels a kernel is slow on."""
import csv, sys, re
f, sub, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(f)) if sub in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-per_step:]
for r in rows:
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void rln::', '')
    print(f"{name[:44]:44s} grid {str(g):16s} {d:8.1f} us  vgpr {r.get('VGPR_Count','?')} lds {r.get('LDS_Block_Size','?')}")
