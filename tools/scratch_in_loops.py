"""Scratch (spill / stack-object) accesses of every kernel in an AMDGPU assembly listing, with the loop depth of the basic
block each sits in.  Scratch shares the vmcnt counter with global loads: a reload inside a pipelined loop is a
vmcnt(0)-class wait that drains the loads in flight.  usage: hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S x.hip -o x.s;
python tools/scratch_in_loops.py x.s [name filter]"""
import re
import sys
import subprocess

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
kern, depth, stats = None, 0, {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern, depth = m.group(1), 0
        stats[kern] = {}
        continue
    if kern is None:
        continue
    if line.startswith(".LBB") or line.startswith("; %bb."):
        d = re.search(r"Depth=(\d+)", line)
        depth = int(d.group(1)) if d else 0
    if "s_endpgm" in line:
        kern = None
        continue
    m = re.search(r"\b(scratch_load|scratch_store)", line)
    if m:
        key = (m.group(1), depth)
        stats[kern][key] = stats[kern].get(key, 0) + 1
for k, v in stats.items():
    if not v or flt not in k:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", k], capture_output=True, text=True).stdout.strip()
    except Exception:  # noqa: BLE001
        name = k
    print(name[:90])
    for (op, d), n in sorted(v.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"    depth {d}: {op} x{n}")
