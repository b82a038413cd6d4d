"""TransitionUp kernels on the 16-bit MFMA pipe with split fp32 operands (csrc/ct3.*) through the C ABI against plain
PyTorch CPU operators (fp64 reference, tolerance per arithmetic mode as in test_gpu_dense3.py)."""
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

# the five TransitionUp geometries of FCDenseNet67 at 120x160 plus ragged crops (odd output width / height)
GEOMS = [(3, 5, 7, 10), (7, 10, 15, 20), (15, 20, 30, 40), (30, 40, 60, 80), (60, 80, 120, 160), (8, 8, 16, 16),
         (8, 8, 17, 17), (5, 3, 10, 7), (4, 6, 9, 12)]


@pytest.mark.parametrize("h,w,ho,wo", GEOMS)
@pytest.mark.parametrize("cin,cout", [(80, 80), (48, 24), (192, 64)])
@pytest.mark.parametrize("parts,dtype", [(2, 1), (2, 0), (3, 0), (1, 0)])
def test_tu_forward(h, w, ho, wo, cin, cout, parts, dtype):
    if (cin, cout) != (80, 80) and ((h, w) in [(60, 80), (30, 40)] or (parts, dtype) in [(3, 0), (1, 0)]):
        pytest.skip("secondary channel counts run on the small geometries in the two-part modes")
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 100 + w + cin + parts)
    n, x_ctot, x_coff, out_ctot, out_coff = 2, cin + 8, 8, cout + 6, 2
    x = torch.randn(n, x_ctot, h, w, generator=g)
    wt = torch.randn(cin, cout, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv_transpose2d(x[:, x_coff:x_coff + cin].double(), wt.double(), bias.double(), stride=2)[:, :, :ho, :wo]
    dev = "cuda"
    out = torch.full((n, out_ctot, ho, wo), 3.0, device=dev)
    stats = torch.zeros(cout, 2, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    xd, wd, bd = x.to(dev), wt.to(dev), bias.to(dev)
    L.check(lib.rln_op_tu_fwd(_p(xd), n, cin, x_ctot, x_coff, h, w, _p(wd), _p(bd), cout, _p(out), out_ctot, out_coff, ho,
                              wo, _p(stats), parts, dtype, _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    sel = got[:, out_coff:out_coff + cout].double()
    err = float((sel - ref).abs().max()) / float(ref.abs().max())
    assert err < TOL[(parts, dtype)], err
    assert torch.all(got[:, :out_coff] == 3.0) and torch.all(got[:, out_coff + cout:] == 3.0)
    s = stats.cpu().double()
    assert torch.allclose(s[:, 0], sel.sum((0, 2, 3)), atol=1e-3, rtol=1e-4)
    assert torch.allclose(s[:, 1], (sel * sel).sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


BWD_GEOMS = [(3, 5, 7, 10), (7, 10, 15, 20), (15, 20, 30, 40), (8, 8, 16, 16), (30, 40, 60, 80), (16, 24, 33, 48),
             (60, 80, 120, 160)]


@pytest.mark.parametrize("h,w,ho,wo", BWD_GEOMS)
@pytest.mark.parametrize("cin,cout", [(80, 80), (48, 24), (192, 64)])
@pytest.mark.parametrize("parts,dtype", [(2, 0), (2, 1), (3, 0)])
def test_tu_backward(h, w, ho, wo, cin, cout, parts, dtype):
    if (cin, cout) != (80, 80) and ((h, w) in [(60, 80), (30, 40)] or (parts, dtype) != (2, 0)):
        pytest.skip("secondary channel counts run on the small geometries in the default backward mode")
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 100 + w + cin + parts)
    n = 2
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn(cin, cout, 3, 3, generator=g, dtype=torch.float64) / (3 * cin ** 0.5)).requires_grad_()
    du = torch.randn(n, cout, ho, wo, generator=g, dtype=torch.float64)
    cscale = torch.rand(cin, generator=g) + 0.5
    out = F.conv_transpose2d(x, wt, None, stride=2)[:, :, :ho, :wo]
    out.backward(du)
    ref_dx = x.grad * cscale.double()[None, :, None, None]
    ref_dw = wt.grad
    dev = "cuda"
    xd, wd, dud, csd = x.detach().float().to(dev), wt.detach().float().to(dev), du.float().to(dev), cscale.to(dev)
    dx = torch.full((n, cin, h, w), 5.0, device=dev)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    want_dw = w % 8 == 0
    dw = torch.full((cin, cout, 3, 3), 5.0, device=dev) if want_dw else None
    L.check(lib.rln_op_tu_bwd(_p(xd), _p(dud), _p(wd), n, cin, cout, h, w, ho, wo, _p(csd), _p(dx), _p(dw), parts, dtype,
                              _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    tol = 4 * TOL[(parts, dtype)]   # sums of up to 9 * cout (dx) / N*H*W (dw) products
    err = float((dx.cpu().double() - ref_dx).abs().max()) / float(ref_dx.abs().max())
    assert err < tol, ("dx", err)
    if want_dw:
        err = float((dw.cpu().double() - ref_dw).abs().max()) / float(ref_dw.abs().max())
        assert err < tol, ("dw", err)
