"""Compressed view of the memory pipeline of one kernel in an AMDGPU listing: loop headers, barriers, groups of global
loads / stores and every s_waitcnt vmcnt(N), in program order -- to see at a glance whether a pipelined loop keeps its
loads in flight (vmcnt(N > 0)) or drains them (vmcnt(0)).  usage: python tools/loop_waits.py x.s <mangled-name substring>"""
import re
import sys

path, flt = sys.argv[1], sys.argv[2]
on = False
ld = st = 0
out = []


def flush():
    global ld, st
    if ld:
        out.append(f"      [{ld} loads]")
    if st:
        out.append(f"      [{st} stores]")
    ld = st = 0


for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        on = flt in m.group(1)
        if on:
            out.append(m.group(1))
        continue
    if not on:
        continue
    if "s_endpgm" in line:
        flush()
        on = False
        continue
    if re.search(r"\b(global_load|buffer_load|scratch_load)", line):
        ld += 1
        continue
    if re.search(r"\b(global_store|buffer_store|scratch_store)", line):
        st += 1
        continue
    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
    if m:
        flush()
        out.append(f"      wait vmcnt({m.group(1)})")
        continue
    if "s_barrier" in line:
        flush()
        out.append("   -- barrier")
        continue
    m = re.match(r"^(\.LBB\w+):.*(=>\s*This.*Loop Header: Depth=(\d+))", line)
    if m:
        flush()
        out.append(f" LOOP {m.group(1)} depth {m.group(3)}")
# collapse runs of decreasing waits
res, run = [], []
for o in out:
    if o.startswith("      wait"):
        run.append(o.strip()[5:])
        continue
    if run:
        res.append("      wait " + (", ".join(run) if len(run) <= 4 else f"{run[0]} .. {run[-1]} ({len(run)} waits)"))
        run = []
    res.append(o)
if run:
    res.append("      wait " + ", ".join(run))
print("\n".join(res))
