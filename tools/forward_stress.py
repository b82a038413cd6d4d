"""Bitwise repeatability of the whole eval forward (and of the training step) in steady state: the same call N times on one
workspace, every result compared with the first.  usage: python tools/forward_stress.py [eval_iters] [train_iters] [bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fcdensenet_oracle as O  # noqa: E402  (initialiser only)
from sim2real_lane_segment_amd.engine import Engine, NetSpec  # noqa: E402

nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
ne, nt = (nums + [2000, 200])[:2]
cfg = O.fcdensenet67_config(4)
st = O.init_state(cfg, 21)
arith = [a for a in sys.argv[1:] if "," in a]
if "exact" in sys.argv or arith:
    from sim2real_lane_segment_amd.engine import parse_dense_arith
    eng = Engine(NetSpec(n_classes=4), device="cuda", dense_arith=parse_dense_arith(arith[0] if arith else "fp32,fp32"))
else:
    eng = Engine(NetSpec(n_classes=4), device="cuda")
if "bf16" in sys.argv:
    eng.set_storage("bf16")
eng.load_state(st)
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 3, 120, 160, generator=g).cuda()
y = torch.randint(0, 4, (64, 120, 160), generator=g).cuda()
ref = eng.forward(x, training=False)[0].clone()
bad = 0
for it in range(ne):
    p = eng.forward(x, training=False)[0]
    if not torch.equal(p, ref):
        d = (p - ref).abs()
        sa = torch.nonzero(d.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
        r_ = torch.nonzero(d[sa[0]].amax(dim=(0, 2)) > 0).flatten().tolist()
        print(f"  eval iteration {it}: max {float(d.max()):.3e}, samples {sa[:16]} ({len(sa)}), sample {sa[0]} rows {r_[:3]}..{r_[-2:]} "
              f"({len(r_)}), pixels {int((d[sa[0]].amax(0) > 0).sum())}")
        bad += 1
print(f"eval forward: {bad} of {ne} repeats differ")
refg = None
badt = 0
for it in range(nt):
    eng.load_state(st)
    probs, _ = eng.forward(x, training=True, with_backward=True, seed=77)
    out, _, _ = eng.loss(probs, y, weighted=True)
    eng.backward(1.0)
    cur = (probs.clone(), eng.grads.clone())
    if refg is None:
        refg = cur
    elif not (torch.equal(cur[0], refg[0]) and torch.equal(cur[1], refg[1])):
        print(f"  train iteration {it}: probs equal {torch.equal(cur[0], refg[0])}, grads max diff {float((cur[1] - refg[1]).abs().max()):.3e}")
        badt += 1
print(f"training step: {badt} of {nt - 1} repeats differ")
print("forward stress:", "DIFFERENCES SEEN" if bad + badt else "clean")
