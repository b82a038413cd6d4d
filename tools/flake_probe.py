"""Repeats the bitwise-reproducibility check of tests/test_gpu_parity.py::test_full_size_properties_batch64 in one process
and reports which tensors differ between two identical training steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import make_engine, O
cfg = O.fcdensenet67_config(4)
st = O.init_state(cfg, 21)
eng = make_engine(cfg, st)
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    eng.set_storage("bf16")
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 3, 120, 160, generator=g).cuda()
y = torch.randint(0, 4, (64, 120, 160), generator=g).cuda()
ref = None
bad = 0
for it in range(12):
    eng.load_state(st)
    probs, _ = eng.forward(x, training=True, with_backward=True, seed=77)
    out, _, _ = eng.loss(probs, y, weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    cur = (probs.clone(), out.clone(), eng.grads.clone())
    if ref is None:
        ref = cur
        continue
    if not torch.equal(ref[0], cur[0]):
        print(it, "probs differ", float((ref[0] - cur[0]).abs().max())); bad += 1
    if not torch.equal(ref[2], cur[2]):
        bad += 1
        d = (ref[2] - cur[2]).abs()
        names = [m.name for m in eng.metas if m.kind == 0 and float((eng.grad_views[m.name] - ref[2][m.offset:m.offset + m.numel].view_as(eng.grad_views[m.name])).abs().max()) > 0]
        print(it, "grads differ: max", float(d.max()), "n tensors", len(names), names[:12])
print("iterations with a difference:", bad)
# eval: batch == concatenation of its halves, repeated
bad = 0
for it in range(10):
    p_all = eng.forward(x, training=False)[0].clone()
    p_a = eng.forward(x[:32].contiguous(), training=False)[0].clone()
    p_b = eng.forward(x[32:].contiguous(), training=False)[0].clone()
    torch.cuda.synchronize()
    if not (torch.equal(p_all[:32], p_a) and torch.equal(p_all[32:], p_b)):
        bad += 1
        print(it, "eval halves differ", float((p_all[:32] - p_a).abs().max()), float((p_all[32:] - p_b).abs().max()))
print("eval iterations with a difference:", bad)
# linearity
bad = 0
for it in range(10):
    eng.load_state(st)
    probs, _ = eng.forward(x, training=True, with_backward=True, seed=77)
    eng.loss(probs, y, weighted=True)
    eng.backward(1.0); torch.cuda.synchronize(); g1 = eng.grads.clone()
    eng.backward(2.0); torch.cuda.synchronize()
    if not torch.allclose(eng.grads, 2 * g1, rtol=1e-5, atol=1e-12):
        bad += 1
        d = (eng.grads - 2 * g1).abs()
        i = int(d.argmax())
        print(it, "linearity off: max", float(d.max()), "at", i, float(eng.grads[i]), float(2 * g1[i]))
print("linearity iterations off:", bad)
