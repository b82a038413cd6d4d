"""Eval forward of a batch against the forwards of its halves (tests/test_gpu_parity.py::test_full_size_properties_batch64,
first property), with a localisation of any difference: which samples / rows / columns / classes, and which of the
forwards changes when repeated.  usage: python tools/half_probe.py [warm]  (warm: run the preceding parity tests' engine
sizes first, as the test file does)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fcdensenet_oracle as O  # noqa: E402  (initialiser only)
from sim2real_lane_segment_amd.engine import Engine, NetSpec  # noqa: E402

cfg = O.fcdensenet67_config(4)
st = O.init_state(cfg, 21)
if "warm" in sys.argv:
    for n, h, w in ((2, 120, 160), (1, 480, 640), (8, 120, 160)):
        e = Engine(NetSpec(n_classes=4), device="cuda")
        e.load_state(st)
        xx = torch.randn(n, 3, h, w).cuda()
        e.forward(xx, training=False)
        p, _ = e.forward(xx, training=True, with_backward=True, seed=3)
        e.loss(p, torch.randint(0, 4, (n, h, w)).cuda(), weighted=True)
        e.backward(1.0)
        torch.cuda.synchronize()
        del e
eng = Engine(NetSpec(n_classes=4), device="cuda")
eng.load_state(st)
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 3, 120, 160, generator=g).cuda()


def fwd(t):
    return eng.forward(t, training=False)[0].clone()


p_all = fwd(x)
p_a = fwd(x[:32].contiguous())
p_b = fwd(x[32:].contiguous())
ok_a, ok_b = torch.equal(p_all[:32], p_a), torch.equal(p_all[32:], p_b)
print("first half equal", ok_a, " second half equal", ok_b)
if not (ok_a and ok_b):
    for name, full, half in (("first", p_all[:32], p_a), ("second", p_all[32:], p_b)):
        d = (full - half).abs()
        if float(d.max()) == 0:
            continue
        per_s = d.amax(dim=(1, 2, 3))
        bad_s = torch.nonzero(per_s > 0).flatten().tolist()
        print(f"{name} half: max {float(d.max()):.3e}; samples with a difference: {bad_s[:40]} ({len(bad_s)} of 32)")
        s0 = bad_s[0]
        rows = torch.nonzero(d[s0].amax(dim=(0, 2)) > 0).flatten().tolist()
        cols = torch.nonzero(d[s0].amax(dim=(0, 1)) > 0).flatten().tolist()
        print(f"  sample {s0}: rows {rows[:12]}..{rows[-3:]} ({len(rows)}), cols {cols[:12]}..{cols[-3:]} ({len(cols)}), "
              f"pixels {int((d[s0].amax(0) > 0).sum())} of {d.shape[2] * d.shape[3]}")
    p_all2, p_b2, p_a2 = fwd(x), fwd(x[32:].contiguous()), fwd(x[:32].contiguous())
    print("repeat: full == full2", torch.equal(p_all, p_all2), " b == b2", torch.equal(p_b, p_b2), " a == a2",
          torch.equal(p_a, p_a2), " full2 halves == a2 / b2", torch.equal(p_all2[:32], p_a2), torch.equal(p_all2[32:], p_b2))
