"""Times one dense-layer forward launch (op-level entry) on FCDenseNet67 layer shapes.
usage: d3_fwd_bench.py [parts dtype]   (env RLN_D3_DBG selects ablations in a -DRLN_DIAG build)"""
import ctypes, sys, os
import torch
sys.path.insert(0, ".")
from sim2real_lane_segment_amd import _lib
L = _lib.lib()
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def p(t): return None if t is None else ctypes.c_void_p(t.data_ptr())
def run(cin, h, w, n=64, ctot=288, reps=10):
    dev = "cuda"
    x = torch.randn(n, ctot, h, w, device=dev)
    a = torch.rand(cin, device=dev) + 0.5; b = torch.randn(cin, device=dev) * 0.3
    wt = torch.randn(16, cin, 3, 3, device=dev) / (3 * cin ** 0.5); bias = torch.zeros(16, device=dev)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    stats = torch.zeros(16, 2, device=dev)
    def call():
        _lib.check(L.rln_op_dense3_fwd(p(x), n, cin, ctot, 0, h, w, p(a), p(b), p(wt), p(bias), 16, None, p(x), ctot, cin,
                                       p(stats), parts, dt, p(ws), ws.numel(), s))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = 4.0 * n * (cin + 16) * h * w / 1e9
    print(f"cin {cin:4d} {h}x{w}: {ms*1000:8.1f} us/launch (incl. pack+reduce)  {gb/ms:7.1f} GB/s alg  dbg={os.environ.get('RLN_D3_DBG','0')}")
for cin in (48, 112, 272):
    run(cin, 120, 160)
run(352, 60, 80, ctot=368)
run(432, 30, 40, ctot=448)
