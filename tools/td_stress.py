"""Bitwise repeatability of the TransitionDown forward (p1_fwd_k) through rln_op_td_fwd at the level-0 geometry of the
bench (64 x 128 x 120x160 -> 60x80).  usage: python tools/td_stress.py [iters] [parts dtype]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sim2real_lane_segment_amd import _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
parts, dtype = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 1)
lib = L.lib()
dev = "cuda"
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
n, c, h, w = 64, 128, 120, 160
g = torch.Generator().manual_seed(1)
x = torch.randn(n, c, h, w, generator=g).to(dev)
a, b = torch.rand(c, generator=g).add(0.5).to(dev), torch.randn(c, generator=g).mul(0.3).to(dev)
wt = (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5).to(dev)
bias = (torch.randn(c, generator=g) * 0.1).to(dev)
out = torch.zeros(n, c, h // 2, w // 2, device=dev)
idx = torch.zeros(n, c, h // 2, w // 2, dtype=torch.uint8, device=dev)
stats = torch.zeros(c, 2, device=dev)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
ref = None
bad = 0
for it in range(iters):
    L.check(lib.rln_op_td_fwd(P(x), n, c, c, 0, h, w, P(a), P(b), P(wt), P(bias), c, None, P(out), c, 0, P(idx), P(stats), parts,
                              dtype, P(ws), ws.numel(), S()))
    cur = (out.clone(), idx.clone(), stats.clone())
    if ref is None:
        ref = cur
    elif not all(torch.equal(u, v) for u, v in zip(cur, ref)):
        bad += 1
print(f"TD forward parts {parts} dtype {dtype}: {bad} of {iters - 1} repeats differ")
