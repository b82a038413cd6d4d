"""TransitionDown kernels on the 16-bit MFMA pipe with split fp32 operands (csrc/pw1.*) through the C ABI against plain
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

# (n, cin, cout, h, w): the five TransitionDown geometries of FCDenseNet67 at 120x160 scaled down in N, plus ragged ones
CASES = [(2, 128, 128, 24, 32), (2, 208, 208, 12, 16), (2, 288, 288, 30, 40), (3, 368, 368, 15, 20), (5, 448, 448, 7, 10),
         (1, 16, 24, 6, 4), (3, 48, 80, 9, 14), (2, 128, 128, 120, 160)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("parts,dtype", [(2, 1), (2, 0), (3, 0), (1, 0)])
def test_td_forward(case, parts, dtype):
    n, cin, cout, h, w = case
    if (h, w) == (120, 160) and (parts, dtype) != (2, 1):
        pytest.skip("full-size case runs in the default arithmetic")
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 1000 + w + cin + parts)
    x_ctot, x_coff, out_ctot, out_coff = cin + 12, 4, cout + 8, 4
    x = torch.randn(n, x_ctot, h, w, generator=g)
    a = torch.rand(cin, generator=g) + 0.5
    b = torch.randn(cin, generator=g) * 0.3
    wt = torch.randn(cout, cin, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    scale = (torch.rand(n, cout, generator=g) < 0.8).float() * 1.25
    xin = x[:, x_coff:x_coff + cin]
    z = F.relu(xin * a[None, :, None, None] + b[None, :, None, None])
    full = (F.conv2d(z.double(), wt.double()[:, :, None, None], bias.double()) * scale[:, :, None, None].double())
    ref, ref_idx = F.max_pool2d(full, 2, return_indices=True)
    ph, pw = h // 2, w // 2
    dev = "cuda"
    out = torch.full((n, out_ctot, ph, pw), 7.0, device=dev)
    idx = torch.full((n, cout, ph, pw), 9, dtype=torch.uint8, device=dev)
    stats = torch.zeros(cout, 2, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    xd, wd, bd, sd, ad, bbd = (t.to(dev) for t in (x, wt, bias, scale, a, b))
    L.check(lib.rln_op_td_fwd(_p(xd), n, cin, x_ctot, x_coff, h, w, _p(ad), _p(bbd), _p(wd), _p(bd), cout, _p(sd),
                              _p(out), out_ctot, out_coff, _p(idx), _p(stats), parts, dtype, _p(ws), ws.numel(),
                              _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    scale_ref = float(full.abs().max())
    err = float((got[:, out_coff:out_coff + cout].double() - ref).abs().max()) / scale_ref
    assert err < TOL[(parts, dtype)], err
    assert torch.all(got[:, :out_coff] == 7.0) and torch.all(got[:, out_coff + cout:] == 7.0)
    # the stored index points at a window element that is the maximum up to the arithmetic's error
    gi = idx.cpu().long()
    assert int(gi.max()) <= 3
    win = full[:, :, :2 * ph, :2 * pw].reshape(n, cout, ph, 2, pw, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, cout, ph, pw, 4)
    picked = torch.gather(win, 4, gi[..., None])[..., 0]
    assert float((picked - ref).abs().max()) / scale_ref < 4 * TOL[(parts, dtype)]
    if parts >= 2:
        # exact ties aside, the argmax agrees with MaxPool2d's on all but near-ties
        ry, rx = ref_idx // w, ref_idx % w
        ref_pos = (ry % 2) * 2 + (rx % 2)
        assert float((ref_pos == gi).double().mean()) > 0.999
    s = stats.cpu().double()
    gsel = got[:, out_coff:out_coff + cout].double()
    assert torch.allclose(s[:, 0], gsel.sum((0, 2, 3)), atol=1e-3, rtol=1e-4)
    assert torch.allclose(s[:, 1], (gsel * gsel).sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


def test_td_forward_unsupported_geometry_is_reported():
    L, lib = _lib()
    x = torch.zeros(1, 16, 6, 5, device="cuda")   # odd width
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    o = torch.zeros(1, 16, 3, 2, device="cuda")
    i8 = torch.zeros(1, 16, 3, 2, dtype=torch.uint8, device="cuda")
    v = torch.zeros(16 * 16, device="cuda")
    rc = lib.rln_op_td_fwd(_p(x), 1, 16, 16, 0, 6, 5, _p(v), _p(v), _p(v), _p(v), 16, None, _p(o), 16, 0, _p(i8), None, 2, 0,
                           _p(ws), ws.numel(), _stream())
    assert rc == -4


BWD_CASES = [(2, 128, 128, 24, 32), (2, 208, 208, 12, 16), (3, 368, 368, 15, 20), (5, 448, 448, 7, 10), (1, 16, 24, 6, 4),
             (3, 48, 80, 9, 14), (2, 128, 128, 120, 160)]


@pytest.mark.parametrize("case", BWD_CASES)
@pytest.mark.parametrize("parts,dtype", [(2, 0), (2, 1), (3, 0)])
def test_td_backward(case, parts, dtype):
    """Data gradient (ReLU mask, gamma scaling, accumulate / overwrite ranges, BatchNorm-backward sums) and weight gradient
    against fp64 autograd of relu(a x + b) -> conv1x1 -> max_pool2d, with the pooled gradient and the forward's argmax
    bytes as the kernels' inputs."""
    n, cin, cout, h, w = case
    if (h, w) == (120, 160) and (parts, dtype) != (2, 0):
        pytest.skip("full-size case runs in the default backward arithmetic")
    L, lib = _lib()
    gen = torch.Generator().manual_seed(h * 1000 + w + cin + parts)
    x = torch.randn(n, cin, h, w, generator=gen)
    a = torch.rand(cin, generator=gen) + 0.5
    b = torch.randn(cin, generator=gen) * 0.3
    gamma = torch.rand(cin, generator=gen) + 0.5
    mean = torch.randn(cin, generator=gen) * 0.1
    invstd = torch.rand(cin, generator=gen) + 0.5
    wt = torch.randn(cout, cin, generator=gen) / cin ** 0.5
    ph, pw = h // 2, w // 2
    dyp = torch.randn(n, cout, ph, pw, generator=gen)
    g0 = torch.randn(n, cin, h, w, generator=gen)
    acc_lo, acc_hi = cin // 4, cin // 2
    # fp64 reference
    xd = x.double()
    z = F.relu(xd * a.double()[None, :, None, None] + b.double()[None, :, None, None]).requires_grad_()
    wd = wt.double().requires_grad_()
    full = F.conv2d(z, wd[:, :, None, None])
    pooled, idx = F.max_pool2d(full, 2, return_indices=True)
    pooled.backward(dyp.double())
    gz = z.grad * (z > 0)
    ref_g = gamma.double()[None, :, None, None] * gz
    accm = torch.zeros(cin, dtype=torch.bool)
    accm[acc_lo:acc_hi] = True
    ref_g = ref_g + g0.double() * accm[None, :, None, None]
    xh = (xd - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    ref_stats = torch.stack([gz.sum((0, 2, 3)), (gz * xh).sum((0, 2, 3))], 1)
    ref_dw = wd.grad
    iy, ix = idx // w - 2 * torch.arange(ph)[None, None, :, None], idx % w - 2 * torch.arange(pw)
    code = (iy * 2 + ix).to(torch.uint8)
    dev = "cuda"
    t = lambda v: v.to(dev)
    xg, dg, cg, wg, ag, bg, gg, mg, ig = (t(v) for v in (x, dyp, code.contiguous(), wt, a, b, gamma, mean, invstd))
    g = g0.clone().to(dev)
    stats = torch.zeros(cin, 2, device=dev)
    dw = torch.full((cout, cin), 5.0, device=dev)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    L.check(lib.rln_op_td_bwd(_p(xg), _p(dg), _p(cg), _p(wg), n, cin, cout, h, w, _p(ag), _p(bg), _p(gg), _p(mg), _p(ig),
                              acc_lo, acc_hi, _p(g), _p(stats), _p(dw), parts, dtype, _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    tol = 4 * TOL[(parts, dtype)]
    scale_g = float((gamma.double()[None, :, None, None] * gz).abs().max())
    assert float((g.cpu().double() - ref_g).abs().max()) / scale_g < tol
    assert float((dw.cpu().double() - ref_dw).abs().max()) / float(ref_dw.abs().max()) < tol
    assert float((stats.cpu().double() - ref_stats).abs().max()) / float(ref_stats.abs().max()) < tol
