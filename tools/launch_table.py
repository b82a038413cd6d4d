"""Per-launch table of one training step: class, device time, algorithmic GB/s and TFLOP/s of every launch (HIP events
inside librln.so).  usage: launch_table.py [f32|bf16] [class-substring]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sim2real_lane_segment_amd import _lib
from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
from sim2real_lane_segment_amd.synthetic import make_batch
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
want = sys.argv[2] if len(sys.argv) > 2 else "dense3"
m = SimpleTrainModule(num_cls=4).cuda(); m.train(); eng = m._rln_sync()
if mode == "bf16":
    eng.set_storage("bf16")
x, y = make_batch(64, device="cuda")
L = _lib.lib()
def step():
    probs, _ = eng.forward(x, training=True, with_backward=True)
    eng.loss(probs, y, weighted=True)
    eng.backward()
for _ in range(3): step()
torch.cuda.synchronize()
_lib.check(L.rln_profile_enable(eng.ctx, 1), "enable")
step(); torch.cuda.synchronize()
cap = 4096
cls = (ctypes.c_int * cap)(); ms = (ctypes.c_double * cap)(); fl = (ctypes.c_double * cap)(); by = (ctypes.c_double * cap)()
n = L.rln_profile_entries(eng.ctx, cls, ms, fl, by, cap)
_lib.check(L.rln_profile_enable(eng.ctx, 0), "disable")  # (clears the collection)
names = [L.rln_profile_class_name(i).decode() for i in range(L.rln_profile_num_classes())]
for i in range(min(n, cap)):
    nm = names[cls[i]]
    if want in nm and ms[i] > 0:
        print(f"{i:4d} {nm:22s} {ms[i]*1e3:8.1f} us  {by[i]/1e6:8.1f} MB  {by[i]/ms[i]/1e6:7.0f} GB/s  {fl[i]/ms[i]/1e9:7.1f} TFLOP/s")
