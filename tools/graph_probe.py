"""Probe: how much of the step is launch gaps?  Captures one training step (fixed seed / AdamW step: timing only)
in a HIP graph through torch.cuda.CUDAGraph and compares replay with eager launches."""
import sys, time, torch
sys.path.insert(0, ".")
from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
from sim2real_lane_segment_amd.synthetic import make_batch
from sim2real_lane_segment_amd.trainer import TrainStepper
m = SimpleTrainModule(num_cls=4).cuda(); m.train(); eng = m._rln_sync()
x, y = make_batch(64, device="cuda")
st = TrainStepper(eng)
def step():
    return st.step(x, y, seed=7)
for _ in range(3): step()
torch.cuda.synchronize()
def timeit(f, n=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
print("graph ms/step", timeit(g.replay))
print("eager again  ", timeit(step))
