"""Bitwise repeatability of the TransitionUp forward (c3_fwd_k) and of its statistics through rln_op_tu_fwd: the level-0
geometry of the bench (64 x 80 x 60x80 -> 120x160), the same call repeated, outputs and [cout, 2] sums compared with the
first.  usage: python tools/tu_stress.py [iters] [parts dtype]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sim2real_lane_segment_amd import _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
parts, dtype = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 1)
lib = L.lib()
dev = "cuda"
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
n, cin, cout, h, w = 64, 80, 80, 60, 80
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g).to(dev)
wt = (torch.randn(cin, cout, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
out = torch.zeros(n, cout, 2 * h, 2 * w, device=dev)
stats = torch.zeros(cout, 2, device=dev)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
ref_o = ref_s = None
bad_o = bad_s = 0
chan = {}
for it in range(iters):
    L.check(lib.rln_op_tu_fwd(P(x), n, cin, cin, 0, h, w, P(wt), P(bias), cout, P(out), cout, 0, 2 * h, 2 * w, P(stats), parts, dtype,
                              P(ws), ws.numel(), S()))
    o, s_ = out.clone(), stats.clone()
    if ref_o is None:
        ref_o, ref_s = o, s_
        continue
    if not torch.equal(o, ref_o):
        bad_o += 1
    if not torch.equal(s_, ref_s):
        bad_s += 1
        for c_ in torch.nonzero((s_ != ref_s).any(dim=1)).flatten().tolist():
            chan[c_] = chan.get(c_, 0) + 1
print(f"TU forward parts {parts} dtype {dtype}: outputs differ in {bad_o}, statistics in {bad_s} of {iters - 1} repeats; "
      f"channels whose sums changed: {dict(sorted(chan.items()))}")
