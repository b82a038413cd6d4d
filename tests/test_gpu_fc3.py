"""First-convolution kernels on the 16-bit MFMA pipe with split fp32 operands (csrc/fc3.*) through the C ABI against
PyTorch CPU operators in fp64 (tolerance per arithmetic mode as in test_gpu_dense3.py)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from sim2real_lane_segment_amd import _lib as L
    return L, L.lib()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


TOL = {(1, 0): 2e-2, (2, 0): 1e-4, (3, 0): 3e-6, (1, 1): 3e-3, (2, 1): 3e-6}
GEOMS = [(2, 3, 48, 120, 160), (3, 1, 48, 9, 16), (2, 3, 40, 15, 24), (1, 2, 64, 33, 8), (5, 3, 48, 7, 40)]


@pytest.mark.parametrize("n,cin,cout,h,w", GEOMS)
@pytest.mark.parametrize("parts,dtype", [(2, 1), (2, 0), (3, 0), (1, 0)])
def test_first_conv_forward(n, cin, cout, h, w, parts, dtype):
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 100 + w + cin + parts)
    out_ctot, out_coff = cout + 8, 4
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    dev = "cuda"
    out = torch.full((n, out_ctot, h, w), 3.0, device=dev)
    stats = torch.zeros(cout, 2, device=dev)
    ws = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
    xd, wd, bd = x.to(dev), wt.to(dev), bias.to(dev)
    L.check(lib.rln_op_fc_fwd(_p(xd), n, cin, h, w, _p(wd), _p(bd), cout, _p(out), out_ctot, out_coff, _p(stats), parts,
                              dtype, _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    sel = got[:, out_coff:out_coff + cout].double()
    assert float((sel - ref).abs().max()) / float(ref.abs().max()) < TOL[(parts, dtype)]
    assert torch.all(got[:, :out_coff] == 3.0) and torch.all(got[:, out_coff + cout:] == 3.0)
    s = stats.cpu().double()
    assert torch.allclose(s[:, 0], sel.sum((0, 2, 3)), atol=1e-3, rtol=1e-4)
    assert torch.allclose(s[:, 1], (sel * sel).sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("n,cin,cout,h,w", GEOMS)
@pytest.mark.parametrize("parts,dtype", [(2, 0), (2, 1), (3, 0)])
def test_first_conv_weight_gradient(n, cin, cout, h, w, parts, dtype):
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 100 + w + cin + parts + 7)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, padding=1).backward(dy.double())
    ref = wt.grad
    dev = "cuda"
    dw = torch.full((cout, cin, 3, 3), 5.0, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    xd, dyd = x.to(dev), dy.to(dev)
    L.check(lib.rln_op_fc_wgrad(_p(xd), _p(dyd), n, cin, cout, h, w, _p(dw), parts, dtype, _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    assert float((dw.cpu().double() - ref).abs().max()) / float(ref.abs().max()) < 4 * TOL[(parts, dtype)]
