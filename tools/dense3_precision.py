"""Parity study of the split-operand dense kernels: error of one DenseLayer forward (BN+ReLU -> conv3x3) against an
fp64 reference, for the exact-fp32 MFMA kernel and every (parts, dtype) mode of csrc/dense3, on layer shapes of
FCDenseNet67 level 0/1 with N(0,1) stacks and Kaiming-uniform weights (the reference's default init).  Run on the GPU
box: python tools/dense3_precision.py > gpurun_out/dense3_precision.txt"""
import ctypes
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from sim2real_lane_segment_amd import _lib  # noqa: E402

L = _lib.lib()


def p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def run(cin, h, w, n=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    a = torch.rand(cin, generator=g) + 0.5
    b = torch.randn(cin, generator=g) * 0.3
    bound = 1.0 / (cin * 9) ** 0.5
    wt = (torch.rand(16, cin, 3, 3, generator=g) * 2 - 1) * bound
    bias = (torch.rand(16, generator=g) * 2 - 1) * bound
    z = F.relu(x.double() * a[None, :, None, None].double() + b[None, :, None, None].double())
    ref = F.conv2d(z, wt.double(), bias.double(), padding=1)
    ref32 = F.conv2d(F.relu(x * a[None, :, None, None] + b[None, :, None, None]), wt, bias, padding=1)
    dev = "cuda"
    xd, ad, bd, wd, biasd = (t.to(dev) for t in (x, a, b, wt, bias))
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = {}
    out = torch.zeros(n, 16, h, w, device=dev)
    _lib.check(L.rln_op_conv_bnrelu(p(xd), n, cin, cin, 0, h, w, p(ad), p(bd), p(wd), p(biasd), 16, 3, None, p(out), 16, 0,
                                    0, None, None, p(ws), ws.numel(), s))
    torch.cuda.synchronize()
    rows["fp32_mfma_kernel"] = out.cpu()
    rows["torch_cpu_fp32"] = ref32
    for parts, dt, name in [(1, 0, "bf16x1"), (2, 0, "bf16x2"), (3, 0, "bf16x3"), (1, 1, "f16x1"), (2, 1, "f16x2")]:
        out = torch.zeros(n, 16, h, w, device=dev)
        _lib.check(L.rln_op_dense3_fwd(p(xd), n, cin, cin, 0, h, w, p(ad), p(bd), p(wd), p(biasd), 16, None, p(out), 16, 0,
                                       None, parts, dt, p(ws), ws.numel(), s))
        torch.cuda.synchronize()
        rows[name] = out.cpu()
    scale = float(ref.abs().max())
    res = {}
    for k, v in rows.items():
        d = (v.double() - ref).abs()
        res[k] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "max_rel_to_max": float(d.max()) / scale}
    return {"cin": cin, "h": h, "w": w, "ref_absmax": scale, "ref_absmean": float(ref.abs().mean()), "modes": res}


if __name__ == "__main__":
    outs = [run(48, 120, 160), run(272, 120, 160), run(128, 60, 80), run(352, 60, 80), run(432, 30, 40)]
    for o in outs:
        print(json.dumps(o))
