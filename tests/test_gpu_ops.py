"""Single-kernel parity (-m gpu): each HIP kernel family of librln.so, called through the C ABI,
against plain PyTorch fp32 CPU operators on the same seeded inputs."""
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


SIZES = [(16, 32), (30, 40), (15, 20), (7, 10), (3, 5), (33, 47), (24, 64)]


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("cin,cout,raw", [(24, 16, False), (3, 48, True), (52, 12, False)])
def test_conv3x3_bnrelu_store(h, w, cin, cout, raw):
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 1000 + w + cin)
    n, x_ctot, x_coff, out_ctot, out_coff = 2, cin + 11, 5, cout + 9, 4
    x = torch.randn(n, x_ctot, h, w, generator=g)
    a = torch.rand(cin, generator=g) + 0.5
    b = torch.randn(cin, generator=g) * 0.3
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    scale = (torch.rand(n, cout, generator=g) < 0.8).float() * 1.25
    xin = x[:, x_coff:x_coff + cin]
    z = xin if raw else F.relu(xin * a[None, :, None, None] + b[None, :, None, None])
    ref = F.conv2d(z, wt, bias, padding=1) * scale[:, :, None, None]

    dev = "cuda"
    out = torch.full((n, out_ctot, h, w), 7.0, device=dev)
    stats = torch.zeros(cout, 2, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    xd, wd, bd, sd = x.to(dev), wt.to(dev), bias.to(dev), scale.to(dev)
    ad, bbd = (None, None) if raw else (a.to(dev), b.to(dev))
    L.check(lib.rln_op_conv_bnrelu(_p(xd), n, cin, x_ctot, x_coff, h, w, _p(ad), _p(bbd), _p(wd), _p(bd), cout, 3,
                                   _p(sd), _p(out), out_ctot, out_coff, 0, None, _p(stats), _p(ws), ws.numel(),
                                   _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.allclose(got[:, out_coff:out_coff + cout], ref, atol=tol, rtol=1e-5)
    # untouched channels keep their fill value
    assert torch.all(got[:, :out_coff] == 7.0) and torch.all(got[:, out_coff + cout:] == 7.0)
    s = stats.cpu()
    assert torch.allclose(s[:, 0], ref.sum((0, 2, 3)), atol=1e-3, rtol=1e-4)
    assert torch.allclose(s[:, 1], (ref * ref).sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("h,w", [(30, 40), (15, 20), (7, 10), (16, 32), (60, 80)])
@pytest.mark.parametrize("c", [40, 128])
def test_conv1x1_pool(h, w, c):
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 100 + w + c)
    n = 2
    x = torch.randn(n, c, h, w, generator=g)
    a = torch.rand(c, generator=g) + 0.5
    b = torch.randn(c, generator=g) * 0.3
    wt = torch.randn(c, c, 1, 1, generator=g) / c ** 0.5
    bias = torch.randn(c, generator=g) * 0.1
    scale = (torch.rand(n, c, generator=g) < 0.8).float() * 1.25
    pre = F.conv2d(F.relu(x * a[None, :, None, None] + b[None, :, None, None]), wt, bias) * scale[:, :, None, None]
    ref, ref_idx = F.max_pool2d(pre, 2, return_indices=True)
    hp, wp = h // 2, w // 2
    dev = "cuda"
    out = torch.zeros(n, c, hp, wp, device=dev)
    idx = torch.zeros(n, c, hp, wp, dtype=torch.uint8, device=dev)
    stats = torch.zeros(c, 2, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    xd, ad, bd, wd, biasd, sd = (t.to(dev) for t in (x, a, b, wt, bias, scale))  # keep alive across the launch
    L.check(lib.rln_op_conv_bnrelu(_p(xd), n, c, c, 0, h, w, _p(ad), _p(bd), _p(wd), _p(biasd), c, 1, _p(sd), _p(out),
                                   c, 0, 1, _p(idx), _p(stats), _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5)
    # argmax index: compare where the window has a unique maximum with a margin
    iy, ix = ref_idx // w - 2 * torch.arange(hp)[None, None, :, None], ref_idx % w - 2 * torch.arange(wp)
    ref_code = (iy * 2 + ix).to(torch.uint8)
    win = F.unfold(pre.reshape(n * c, 1, h, w)[:, :, :2 * hp, :2 * wp], 2, stride=2).reshape(n, c, 4, hp, wp)
    top2 = win.topk(2, dim=2)[0]
    clear = (top2[:, :, 0] - top2[:, :, 1]) > 1e-4
    assert torch.equal(idx.cpu()[clear], ref_code[clear])
    s = stats.cpu()
    assert torch.allclose(s[:, 0], ref.sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("h,w,ho,wo", [(3, 5, 7, 10), (7, 10, 15, 20), (15, 20, 30, 40), (8, 8, 16, 16),
                                       (8, 8, 17, 17), (30, 40, 60, 80), (5, 3, 10, 7)])
@pytest.mark.parametrize("c", [80, 20])
def test_conv_transpose_crop(h, w, ho, wo, c):
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 100 + w + c)
    n, out_ctot, out_coff = 2, c + 6, 0
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5)
    bias = torch.randn(c, generator=g) * 0.1
    ref = F.conv_transpose2d(x, wt, bias, stride=2)[:, :, :ho, :wo]
    dev = "cuda"
    out = torch.full((n, out_ctot, ho, wo), 3.0, device=dev)
    xd, wd, bd = x.to(dev), wt.to(dev), bias.to(dev)  # keep alive across the launch
    L.check(lib.rln_op_convt(_p(xd), n, c, h, w, _p(wd), _p(bd), c, _p(out), out_ctot, out_coff, ho, wo, _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.allclose(got[:, :c], ref, atol=3e-5, rtol=1e-5)
    assert torch.all(got[:, c:] == 3.0)


@pytest.mark.parametrize("ncls", [2, 4, 12])
def test_classifier_op(ncls):
    from sim2real_lane_segment_amd.engine import classifier_op
    g = torch.Generator().manual_seed(ncls)
    feat = F.normalize(torch.randn(2, 288, 9, 13, generator=g))
    wt = torch.randn(ncls, 288, 1, 1, generator=g) / 17
    b = torch.randn(ncls, generator=g) * 0.1
    ref_l = F.conv2d(feat, wt, b) / 0.05
    got_l = classifier_op(feat.cuda(), wt.cuda(), b.cuda(), 0.05, use_softmax=False).cpu()
    got_p = classifier_op(feat.cuda(), wt.cuda(), b.cuda(), 0.05, use_softmax=True).cpu()
    assert torch.allclose(got_l, ref_l, atol=2e-5, rtol=1e-5)
    assert torch.allclose(got_p, F.softmax(ref_l, 1), atol=2e-6)


@pytest.mark.parametrize("ks", [3, 7])
def test_wide_classifier_op_and_fcdensenet57(ks):
    """FCDenseNetClassifier(kernel_size=k != 1) (tiramisu.py:113-115; FCDenseNet57(n_classes, kernel_size), :150): the
    k x k finalConv + / T + softmax against torch, stand-alone and behind the fused feature extractor."""
    from sim2real_lane_segment_amd.engine import classifier_op
    g = torch.Generator().manual_seed(ks)
    feat = F.normalize(torch.randn(2, 48, 17, 24, generator=g))
    wt = torch.randn(4, 48, ks, ks, generator=g) / (ks * 7)
    b = torch.randn(4, generator=g) * 0.1
    ref_l = F.conv2d(feat, wt, b, padding=ks // 2) / 0.05
    got_l = classifier_op(feat.cuda(), wt.cuda(), b.cuda(), 0.05, use_softmax=False).cpu()
    got_p = classifier_op(feat.cuda(), wt.cuda(), b.cuda(), 0.05, use_softmax=True).cpu()
    assert torch.allclose(got_l, ref_l, atol=5e-5, rtol=1e-5)
    assert torch.allclose(got_p, F.softmax(ref_l, 1), atol=5e-6)
    if ks == 3:
        from sim2real_lane_segment_amd.models.FCDenseNet.tiramisu import FCDenseNet57
        torch.manual_seed(1)
        net = FCDenseNet57(4, kernel_size=3).cuda().eval()
        x = torch.randn(1, 3, 64, 96, generator=g).cuda()
        out = net(x)
        with torch.no_grad():
            feat57 = net.featureExtractor(x)
            ref = F.softmax(F.conv2d(feat57, net.classifier.finalConv.weight, net.classifier.finalConv.bias,
                                     padding=1) / net.classifier.T, 1)
        assert out.shape == (1, 4, 64, 96) and torch.allclose(out, ref, atol=2e-5)
        assert torch.allclose(net.classifier(feat57), ref, atol=2e-5)
        net.train()
        with pytest.raises(RuntimeError, match="inference-only"):
            net(x)


def test_adamw_matches_golden(golden_dir):
    import numpy as np
    L, lib = _lib()
    z = np.load(golden_dir + "/misc.npz")
    p = torch.from_numpy(z["adamw_p0"].copy()).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        gr = torch.from_numpy(z["adamw_grads"][s]).cuda()
        L.check(lib.rln_adamw_step(_p(p), _p(gr), _p(m), _p(v), p.numel(), 1e-3, 0.9, 0.999, 1e-8, 1e-4, s + 1, 1.0,
                                   _stream()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), z[f"adamw_p{s + 1}"], rtol=2e-6, atol=1e-7)
