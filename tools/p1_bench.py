"""Runs the TransitionDown forward op (csrc/pw1.*) at the bench geometry; meant to run under
`rocprofv3 --kernel-trace` (diagnostic builds read RLN_P1_DBG: 1 no loads, 2 no MFMA, 4 no epilogue)."""
import ctypes, sys
import torch
from sim2real_lane_segment_amd import _lib as L

lib = L.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
n, c, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 128, 120, 160)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = "cuda"
x = torch.randn(n, c, h, w, device=dev)
a = torch.rand(c, device=dev) + 0.5
b = torch.randn(c, device=dev) * 0.3
wt = torch.randn(c, c, device=dev) / c ** 0.5
bias = torch.randn(c, device=dev) * 0.1
scale = (torch.rand(n, c, device=dev) < 0.8).float() * 1.25
out = torch.empty(n, c, h // 2, w // 2, device=dev)
idx = torch.empty(n, c, h // 2, w // 2, dtype=torch.uint8, device=dev)
stats = torch.zeros(c, 2, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    L.check(lib.rln_op_td_fwd(P(x), n, c, c, 0, h, w, P(a), P(b), P(wt), P(bias), c, P(scale), P(out), c, 0, P(idx),
                              P(stats), 2, 1, P(ws), ws.numel(), s))
torch.cuda.synchronize()
print("done", float(out.sum()))
