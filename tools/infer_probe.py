"""Inference-path timing (BASELINE.json configs[3]: makeDemoVideo.py, 480x640 frames): eval forward + argmax."""
import sys, time, torch
sys.path.insert(0, ".")
from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
m = SimpleTrainModule(num_cls=4).cuda().eval()
for n, h, w in [(1, 480, 640), (8, 480, 640), (64, 120, 160)]:
    x = torch.randn(n, 3, h, w, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            out = m(x); mask = torch.max(out, 1)[1]
        torch.cuda.synchronize(); t0 = time.perf_counter(); k = 10
        for _ in range(k):
            out = m(x); mask = torch.max(out, 1)[1]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
    print(f"eval forward N={n} {h}x{w}: {dt*1e3:.2f} ms/batch  {n/dt:.1f} images/s")

# HIP-graph replay of the N=1 480x640 eval forward (is the small-batch path launch-bound?)
x = torch.randn(1, 3, 480, 640, device="cuda")
with torch.no_grad():
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); print(f"graph replay N=1 480x640: {(time.perf_counter()-t0)/20*1e3:.2f} ms")
