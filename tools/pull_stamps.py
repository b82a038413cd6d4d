"""Diagnostic (build with -DRLN_DIAG): phase shares of d3_pull_k for the launches with >= RLN_PULL_STAMPS channels."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sim2real_lane_segment_amd import _lib
from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
from sim2real_lane_segment_amd.synthetic import make_batch
m = SimpleTrainModule(num_cls=4).cuda(); m.train(); eng = m._rln_sync()
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    eng.set_storage("bf16")
x, y = make_batch(64, device="cuda")
L = _lib.lib(); buf = (ctypes.c_uint64 * 8)()
for it in range(2):
    probs, _ = eng.forward(x, training=True, with_backward=True)
    eng.loss(probs, y, weighted=True)
    L.rln_debug_read_stamps(buf)  # clears (forward kernels do not write it in this mode)
    eng.backward()
    torch.cuda.synchronize()
    L.rln_debug_read_stamps(buf)
names = ["wait at tile barrier", "dY staging", "wait for staged tile", "item set-up (S loads, addresses)", "MFMA loops",
         "layer epilogues", "G read-modify-write"]
tot = sum(buf[i] for i in range(7)) or 1
for i, n in enumerate(names):
    print(f"{n:34s} {100.0 * buf[i] / tot:5.1f} %")
print("items", buf[7], " wave-cycles per item", tot / max(buf[7], 1))
