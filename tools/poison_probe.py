"""Uninitialised-workspace probe: one training step (and an eval forward) with a fresh workspace as the allocator hands
it out, and again with every workspace byte set to 0xFF (NaN patterns) before the library carves it.  Results must be
finite and bit-identical: a kernel that reads workspace it did not write in the same step would differ here instead of
once in fifty test runs.  usage: python tools/poison_probe.py [bf16] [n h w]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fcdensenet_oracle as O  # noqa: E402  (initialiser only)
from sim2real_lane_segment_amd.engine import Engine, NetSpec  # noqa: E402

args = [a for a in sys.argv[1:]]
bf16 = "bf16" in args
nums = [int(a) for a in args if a.isdigit()]
n, h, w = (nums + [64, 120, 160])[:3] if len(nums) >= 3 else (64, 120, 160)
cfg = O.fcdensenet67_config(4)
st = O.init_state(cfg, 21)
g = torch.Generator().manual_seed(5)
x = torch.randn(n, 3, h, w, generator=g).cuda()
y = torch.randint(0, 4, (n, h, w), generator=g).cuda()


def run(poison):
    if poison:
        os.environ["RLN_POISON_WORKSPACE"] = "1"
    else:
        os.environ.pop("RLN_POISON_WORKSPACE", None)
    eng = Engine(NetSpec(n_classes=4), device="cuda")
    if bf16:
        eng.set_storage("bf16")
    eng.load_state(st)
    pe = eng.forward(x, training=False)[0].clone()
    if not eng.check_ws_guard():
        print("GUARD BROKEN after the eval forward (a kernel wrote past the workspace)")
    half = eng.forward(x[: max(1, n // 2)].contiguous(), training=False)[0].clone()
    if not eng.check_ws_guard():
        print("GUARD BROKEN after the half-batch eval forward")
    if not torch.equal(half, pe[: max(1, n // 2)]):
        print("half-batch eval forward differs from the full batch")
    probs, _ = eng.forward(x, training=True, with_backward=True, seed=77)
    out, _, _ = eng.loss(probs, y, weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    if not eng.check_ws_guard():
        print("GUARD BROKEN after the training step")
    return pe, probs.clone(), out.clone(), eng.grads.clone(), eng


# PROBE_SAVE=file: one unpoisoned run saved; PROBE_CMP=file: one run under the caller's environment (e.g. a diagnostic
# build with RLN_POISON_LDS=1, which is latched per process) compared with the saved one
if os.environ.get("PROBE_SAVE"):
    a = run(False)
    torch.save([t.cpu() for t in a[:4]], os.environ["PROBE_SAVE"])
    print("saved")
    sys.exit(0)
if os.environ.get("PROBE_CMP"):
    a = [t.cuda() for t in torch.load(os.environ["PROBE_CMP"])]
    b = run(bool(os.environ.get("RLN_POISON_WORKSPACE")))
else:
    a = run(False)
    b = run(True)
names = ["eval probs", "train probs", "loss", "grads"]
bad = 0
for i, nm in enumerate(names):
    fin = bool(torch.isfinite(b[i]).all())
    same = bool(torch.equal(a[i], b[i]))
    print(f"{nm:12s} finite {fin}  identical to the unpoisoned run {same}")
    bad += (not fin) + (not same)
if bad:
    eng = b[4]
    gv = eng.grad_views
    ga = a[3]
    for i in range(3):
        if not torch.equal(a[i], b[i]):
            d = (a[i].float() - b[i].float()).abs()
            print(f"  {names[i]}: max abs diff {float(d[torch.isfinite(d)].max()) if torch.isfinite(d).any() else float('nan'):.3e}, "
                  f"non-finite {int((~torch.isfinite(b[i])).sum())}")
    off = [m.name for m in eng.metas if m.kind == 0 and not torch.equal(gv[m.name].reshape(-1), ga[m.offset:m.offset + m.numel])]
    print("gradient tensors that differ:", len(off), off[:16])
print("poison probe:", "FAILED" if bad else "clean")
