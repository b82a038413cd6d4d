"""Throughput of the on-device input transform on frames already resident in HBM (uint8 480x640 -> float 3x120x160)."""
import sys, time, torch
sys.path.insert(0, ".")
from sim2real_lane_segment_amd.dataManagement.myTransforms import MyTransform
n = 64
frames = torch.randint(0, 256, (n, 480, 640, 3), dtype=torch.uint8, device="cuda")
labels = torch.randint(0, 4, (n, 480, 640), dtype=torch.uint8, device="cuda")
for aug in (False, True):
    tf = MyTransform(augment=aug, seed=1)
    for _ in range(3): tf(frames, labels)
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = 20
    for _ in range(k): tf(frames, labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
    print(f"augment={aug}: {dt*1e3:.3f} ms per batch of {n}  ->  {n/dt:,.0f} images/s (host parameter draw included)")
