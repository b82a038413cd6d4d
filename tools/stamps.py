"""Diagnostic: per-phase cycle shares of the dense forward kernel (run with RLN_DBG=16)."""
import ctypes, os, sys, torch
sys.path.insert(0, ".")
from sim2real_lane_segment_amd import _lib
from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
from sim2real_lane_segment_amd.synthetic import make_batch
m = SimpleTrainModule(num_cls=4).cuda(); m.train(); eng = m._rln_sync()
x, y = make_batch(64, device="cuda")
L = _lib.lib(); buf = (ctypes.c_uint64 * 8)()
for it in range(2):
    eng.forward(x, training=True, with_backward=True)
    torch.cuda.synchronize()
    L.rln_debug_read_stamps(buf)
tot = sum(buf[i] for i in range(5)) or 1
names = ["commit(wait loads+LDS write)", "barrier1", "issue loads", "mfma", "barrier2"]
for i, n in enumerate(names):
    print(f"{n:32s} {100.0 * buf[i] / tot:5.1f}%   {buf[i] / max(buf[5],1) / 4:9.0f} cycles per chunk per wave")
print("chunks", buf[5] // 4)
