"""Timing of the paired dense forward (rln_op_dense3_fwd_pair) against two one-layer launches at a bench-size level, with
the diagnostic build's ablation bits (RLN_D3_DBG: 1 no global loads, 2 no conversion / commit, 4 no MFMA phase).
usage: RLN_D3_DBG=.. python tools/pair_abl.py [parts dtype cin n h w]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sim2real_lane_segment_amd import _lib as L  # noqa: E402

parts, dtype, cin, n, h, w = (int(v) for v in (sys.argv[1:] + ["2", "1", "208", "64", "120", "160"][len(sys.argv) - 1:]))
lib = L.lib()
dev = "cuda"
ctot = cin + 32
g = torch.Generator().manual_seed(1)
x = torch.randn(n, ctot, h, w, generator=g).to(dev)
a1, b1 = torch.rand(cin).add(0.5).to(dev), torch.randn(cin).mul(0.3).to(dev)
a2, b2 = torch.rand(cin + 16).add(0.5).to(dev), torch.randn(cin + 16).mul(0.3).to(dev)
w1 = (torch.randn(16, cin, 3, 3) / (3 * cin ** 0.5)).to(dev)
w2 = (torch.randn(16, cin + 16, 3, 3) / (3 * (cin + 16) ** 0.5)).to(dev)
bias = torch.zeros(16, device=dev)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
scratch = torch.empty(n * 16 * h * w, device=dev)
st = torch.zeros(16, 2, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def pair():
    L.check(lib.rln_op_dense3_fwd_pair(P(x), n, cin, ctot, 0, h, w, P(a1), P(b1), P(w1), P(bias), None, P(a2), P(b2), P(w2),
                                       P(bias), None, P(st), P(st), parts, dtype, P(scratch), P(ws), ws.numel(), S()))


def single():
    L.check(lib.rln_op_dense3_fwd(P(x), n, cin, ctot, 0, h, w, P(a1), P(b1), P(w1), P(bias), 16, None, P(x), ctot, cin, P(st),
                                  parts, dtype, P(ws), ws.numel(), S()))
    L.check(lib.rln_op_dense3_fwd(P(x), n, cin + 16, ctot, 0, h, w, P(a2), P(b2), P(w2), P(bias), 16, None, P(x), ctot,
                                  cin + 16, P(st), parts, dtype, P(ws), ws.numel(), S()))


for name, fn in (("two single launches", single), ("pair + finish", pair)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"parts {parts} dtype {dtype} cin {cin} {n}x{h}x{w} dbg {os.environ.get('RLN_D3_DBG', '0')}: {name:22s} "
          f"{(time.perf_counter() - t0) / 10 * 1e6:8.1f} us (incl. weight packing, statistics rows)")
