"""Does re-carving the workspace race with kernels of the previous geometry?  A long device-side sleep is queued in front
of a batch-64 eval forward, then the forward of its second half follows at once (new geometry -> new workspace on the
same memory).  Prints how long the host spent in the second call (a set-up that drains the device first waits for the
sleep) and whether the batch still equals its halves.  usage: python tools/ws_race_probe.py [nosync]
(nosync: the Python-side synchronize is skipped -- with a library that does not drain the device either, the old
behaviour)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fcdensenet_oracle as O  # noqa: E402  (initialiser only)
from sim2real_lane_segment_amd.engine import Engine, NetSpec  # noqa: E402

cfg = O.fcdensenet67_config(4)
st = O.init_state(cfg, 21)
eng = Engine(NetSpec(n_classes=4), device="cuda")
eng.load_state(st)
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 3, 120, 160, generator=g).cuda()
x_copy = x.clone()
ref_b = eng.forward(x[32:].contiguous(), training=False)[0].clone()   # reference result of the second half
ref_all = eng.forward(x, training=False)[0].clone()
torch.cuda.synchronize()
print("quiet run: second half equal", torch.equal(ref_all[32:], ref_b))
big = torch.randn(12288, 12288, device="cuda")
t0 = time.perf_counter()
for _ in range(40):
    big @ big
torch.cuda.synchronize()
print(f"backlog = {1e3 * (time.perf_counter() - t0):.0f} ms of device work")
if "nosync" in sys.argv:
    real_sync = torch.cuda.synchronize
bad = 0
for it in range(int(os.environ.get('ROUNDS', '12'))):
    eng.forward(x[:2].contiguous(), training=False)  # another geometry, so that the batch-64 call re-carves too
    torch.cuda.synchronize()
    for _ in range(40):                               # a few hundred ms of device work in front of the forward
        big @ big
    if "nosync" in sys.argv:
        torch.cuda.synchronize = lambda *a, **k: None
    p_all = eng.forward(x, training=False)[0]
    t0 = time.perf_counter()
    p_b = eng.forward(x[32:].contiguous(), training=False)[0]
    dt = time.perf_counter() - t0
    if "nosync" in sys.argv:
        torch.cuda.synchronize = real_sync
    torch.cuda.synchronize()
    ok = torch.equal(p_all[32:], p_b) and torch.equal(p_all, ref_all)
    bad += not ok
    if not ok:
        da = (p_all - ref_all).abs()
        sa = torch.nonzero(da.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
        if sa:
            r_ = torch.nonzero(da[sa[0]].amax(dim=(0, 2)) > 0).flatten().tolist()
            c_ = torch.nonzero(da[sa[0]].amax(dim=(0, 1)) > 0).flatten().tolist()
            print(f"    batch forward: max diff {float(da.max()):.3e}, samples {sa[:32]} ({len(sa)}); sample {sa[0]}: rows "
                  f"{r_[:3]}..{r_[-2:]} ({len(r_)}), cols {c_[:3]}..{c_[-2:]} ({len(c_)}), pixels {int((da[sa[0]].amax(0) > 0).sum())}; "
                  f"input unchanged {torch.equal(x, x_copy)}")
        p_b2 = eng.forward(x[32:].contiguous(), training=False)[0].clone()   # same workspace, no re-carve
        p_all2 = eng.forward(x, training=False)[0].clone()                   # re-carves again, quiet device
        torch.cuda.synchronize()
        d = (p_b - ref_b).abs()
        bad_s = torch.nonzero(d.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
        print(f"    half-batch: max diff {float(d.max()):.3e}, samples {bad_s[:40]} ({len(bad_s)}), pixels of the first "
              f"{int((d[bad_s[0]].amax(0) > 0).sum()) if bad_s else 0}; repeated on the same workspace == reference "
              f"{torch.equal(p_b2, ref_b)}; batch again == reference {torch.equal(p_all2, ref_all)}")
    print(f"iteration {it}: host time of the half-batch call {dt * 1e3:7.1f} ms; batch == reference {torch.equal(p_all, ref_all)}, "
          f"second half equal {torch.equal(p_all[32:], p_b)}")
print("race probe:", "CORRUPTION SEEN" if bad else "clean")
