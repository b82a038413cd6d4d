"""End-to-end parity (-m gpu): the HIP path through the C ABI against (a) the golden vectors generated
from the reference's own code and (b) the CPU oracle on the same seeded inputs.

Tolerances (fp32 storage / fp32 accumulation, BASELINE.json north_star): argmax masks bit-exact (outside the near-tie
pixels the fixture lists; the actual flip count is printed and bounded), probabilities / scaled logits within 1e-3
absolute, gradients within 1e-3 relative to the tensor's scale on the reduced nets (every tensor, full comparison).

Gradient noise floor on the full nets: the reference algorithm itself, run in fp32 and in fp64 on the CPU with the same
inputs, differs per gradient tensor by an L2-relative median 1.5e-4 / max 2.4e-3 on FCDenseNet67 (2x120x160) and median
1.4e-3 / max 2.6e-2 on FCDenseNet103 (2x64x96), because fp32 rounding flips individual ReLU / max-pool decisions in a
60-100 layer net (tools/grad_noise_floor.py -> tests/golden/grad_noise_floor.json).  Full-net gradients against the
REFERENCE's own training_step (fixture fcd67_train_120x160, 5 steps): exact-fp32 kernels every tensor within
max(3e-3, 5 x its noise floor) and the whole arena within 1e-3; default arithmetic median <= 1e-3, every tensor <= 6e-3
(test_fcd67_train_steps_vs_golden).  Against the oracle run live (other nets / batch sizes): max(6e-3, 5 x floor) capped at
5e-2 (grad_l2_bar).  The forward quantities keep the 1e-3 / bit-exact-argmax bar."""
import os

import numpy as np
import pytest
import torch

from oracle import fcdensenet_oracle as O
from tests.golden.common import cfg_from_arrays, retry_if_not_reproducible, sample_idx, synth_batch, unpack_masks

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"fixture {name} missing")
    return np.load(path)


def make_engine(cfg, st):
    from sim2real_lane_segment_amd.engine import Engine, NetSpec
    spec = NetSpec(in_channels=cfg.in_channels, down_blocks=cfg.down_blocks, up_blocks=cfg.up_blocks,
                   bottleneck_layers=cfg.bottleneck_layers, growth_rate=cfg.growth_rate,
                   out_chans_first_conv=cfg.out_chans_first_conv, n_classes=cfg.n_classes,
                   temperature=cfg.temperature)
    eng = Engine(spec, device="cuda")
    eng.load_state(st)
    return eng


_NOISE = None


def _noise_floor(table, name):
    """The tensor's own fp32-vs-fp64 gradient noise floor: the reference algorithm run by the CPU oracle in fp32 and in
    fp64 on identical inputs (tools/grad_noise_floor.py -> tests/golden/grad_noise_floor.json; FCDenseNet67 2x120x160:
    median 1.5e-4, max 2.4e-3; FCDenseNet103 2x64x96: median 1.4e-3, max 2.6e-2 -- rounding flips individual ReLU /
    max-pool decisions in a 60-100 layer net)."""
    global _NOISE
    if _NOISE is None:
        import json
        with open(os.path.join(GOLDEN, "grad_noise_floor.json")) as f:
            _NOISE = json.load(f)
    return float(_NOISE[table]["per_tensor"].get(name, 0.0))


GRAD_BAR_CAP = 5e-2


def grad_l2_bar(table, name, base=6e-3):
    """Per-tensor bar on the L2-relative gradient error against the oracle: max(base, 5 x the tensor's noise floor),
    never above GRAD_BAR_CAP (tensors whose floor asks for more are listed by the test instead of being waved through).
    base 6e-3: structured frames / larger batches (measured <= 5.2e-3 with the default arithmetic); the random-label
    fixture of test_fcd67_train_steps_vs_golden needs 1.5e-2 and says why."""
    return min(GRAD_BAR_CAP, max(base, 5.0 * _noise_floor(table, name)))


def rel_err(got, ref):
    # floor: a bias in front of BatchNorm-only consumers has an exactly-zero gradient (both sides = rounding noise)
    scale = max(float(np.abs(ref).max()), 1e-5)
    return float(np.abs(got - ref).max()) / scale


@pytest.mark.parametrize("name,full", [("tiny_40x56", True), ("tiny_33x47", True), ("g16_32x48", False),
                                       ("g16_absent_30x34", False)])
def test_small_nets_vs_golden(name, full):
    z = load(name)
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    if int(z["absent_class"]):
        y[y == cfg.n_classes - 1] = 0
    scales = O.make_drop_scales(cfg, n, seed + 2)
    eng = make_engine(cfg, st)

    # ---- eval forward ----
    probs, feat = eng.forward(x.cuda(), training=False, want_feat=True)
    logits, _ = eng.forward(x.cuda(), training=False, use_softmax=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(probs.cpu().numpy(), z["eval_probs"], atol=1e-3)
    np.testing.assert_allclose(logits.cpu().numpy(), z["eval_logits"], atol=1e-3)
    assert np.abs(probs.cpu().numpy() - z["eval_probs"]).max() < 2e-4  # what fp32 actually delivers
    np.testing.assert_allclose(feat.sum((0, 2, 3)).cpu().numpy(), z["eval_feat_sum"], rtol=1e-3, atol=1e-3)
    gap = np.sort(z["eval_logits"], axis=1)
    clear = (gap[:, -1] - gap[:, -2]) > 2e-3
    assert np.array_equal(probs.argmax(1).cpu().numpy()[clear], z["eval_probs"].argmax(1)[clear])

    # ---- training step: forward, loss, backward, AdamW ----
    ds = eng.pack_drop_scales(scales)
    probs_t, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=ds)
    out, _, _ = eng.loss(probs_t, y.cuda(), weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    ref_pt = z["train_probs"]
    got_pt = probs_t.cpu().numpy() if full else probs_t.cpu().numpy()[:, :, ::7, ::5]
    np.testing.assert_allclose(got_pt, ref_pt, atol=1e-3)
    assert abs(float(out[0]) - float(z["train_loss"])) < 1e-4
    assert abs(float(out[1]) * 100 - float(z["train_acc"])) < 0.02
    assert float(out[2]) == 0
    # running statistics after the step
    for k in z.files:
        if k.startswith("buf1/"):
            np.testing.assert_allclose(eng.views[k[5:]].cpu().numpy(), z[k], rtol=1e-4, atol=1e-5, err_msg=k)
    for m in eng.metas:
        if m.name.endswith("num_batches_tracked"):
            assert int(eng.views[m.name]) == 1
    bad = []
    for m in eng.metas:
        if m.kind != 0:
            continue
        g = eng.grad_views[m.name].cpu().numpy()
        if full:
            ref = z["grad/" + m.name]
            err = rel_err(g, ref)
        else:
            idx = sample_idx(g.size, 64, 1234).numpy()
            ref = z["gradsamp/" + m.name]
            nrm = float(z["gradnorm/" + m.name])
            e_samp = float(np.abs(g.reshape(-1)[idx] - ref).max()) / max(float(np.abs(ref).max()),
                                                                         nrm / np.sqrt(g.size), 1e-6)
            e_norm = abs(float(np.linalg.norm(g)) - nrm) / max(nrm, 1e-6 * np.sqrt(g.size))
            err = max(e_samp / 3, e_norm)
        # 3e-3: the deepest levels run split-K / split-channel kernels whose summation order differs from a
        # single pass; on 10x14-pixel levels that alone moves individual weight gradients by ~1.5e-3 (measured)
        if not err < 3e-3:
            bad.append((m.name, err))
    assert not bad, f"gradient mismatches (first in backward order last): {bad[:12]} ... total {len(bad)}"

    grads_gpu = {m.name: eng.grad_views[m.name].cpu().numpy().copy() for m in eng.metas if m.kind == 0}
    params0 = {m.name: eng.views[m.name].cpu().clone() for m in eng.metas if m.kind == 0}
    m_buf = torch.zeros_like(eng.params)
    v_buf = torch.zeros_like(eng.params)
    eng.adamw_step(m_buf, v_buf, 1, 1e-3, weight_decay=1e-4)
    torch.cuda.synchronize()
    for m in eng.metas:
        if m.kind != 0:
            continue
        p = eng.views[m.name].cpu().numpy()
        # (a) exact plumbing check: the oracle's AdamW applied to the GPU's own gradients
        pe = params0[m.name].clone()
        O.adamw_step(pe, torch.from_numpy(grads_gpu[m.name]), torch.zeros_like(pe), torch.zeros_like(pe), 1, 1e-3,
                     weight_decay=1e-4)
        np.testing.assert_allclose(p, pe.numpy(), rtol=1e-5, atol=2e-7, err_msg=m.name)
        # (b) against the reference's updated parameters.  AdamW's first step is lr*g/(|g|+eps): the derivative
        # w.r.t. g is lr*eps/(|g|+eps)^2, so gradient noise on near-zero gradients is amplified up to lr/eps.
        if full:
            gref, got_g, pref, pgot = z["grad/" + m.name], grads_gpu[m.name], z["param1/" + m.name], p
        else:
            idx = sample_idx(p.size, 64, 1234).numpy()
            gref, got_g = z["gradsamp/" + m.name], grads_gpu[m.name].reshape(-1)[idx]
            pref, pgot = z["param1samp/" + m.name], p.reshape(-1)[idx]
        tol = 2e-5 + np.minimum(2.1e-3, 1e-3 * 1e-8 * np.abs(got_g - gref) / (np.abs(gref) + 1e-8) ** 2
                                + 1e-3 * np.abs(got_g - gref) / (np.abs(gref) + 1e-8))
        assert np.all(np.abs(pgot - pref) <= tol + 1e-3 * np.abs(pref)), m.name


@pytest.mark.parametrize("name", ["fcd67_eval_120x160", "fcd67_eval_480x640"])
def test_fcd67_eval_masks_vs_golden(name):
    z = load(name)
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, _ = synth_batch(n, h, w, 4, seed + 1)
    eng = make_engine(cfg, st)
    probs, feat = eng.forward(x.cuda(), training=False, want_feat=True)
    logits, _ = eng.forward(x.cuda(), training=False, use_softmax=False)
    out, am, _ = eng.loss(probs, torch.zeros(n, h, w, dtype=torch.int64).cuda(), weighted=False, want_argmax=True)
    torch.cuda.synchronize()
    mask = am.reshape(-1).cpu()
    assert torch.equal(mask, probs.argmax(1).reshape(-1).cpu())
    ref_mask = unpack_masks(z["mask_packed"], n * h * w)
    near = set(int(i) for i in z["near_tie_idx"])
    diff = torch.nonzero(mask != ref_mask).reshape(-1).tolist()
    print(f"[{name}] argmax flips vs the reference mask: {len(diff)} of {n * h * w} pixels "
          f"({len(near)} pixels have a reference top-2 gap < 2e-3; flips outside them: {sum(d not in near for d in diff)})")
    assert all(d in near for d in diff), f"{len(diff)} argmax flips, {sum(d not in near for d in diff)} outside near ties"
    assert len(diff) == 0, "the kernels are deterministic: the measured flip count on these fixtures is zero"
    idx = torch.from_numpy(z["sample_idx"])
    got_p = probs.permute(0, 2, 3, 1).reshape(-1, 4).cpu()[idx].numpy()
    got_l = logits.permute(0, 2, 3, 1).reshape(-1, 4).cpu()[idx].numpy()
    np.testing.assert_allclose(got_p, z["probs_samp"], atol=1e-3)
    np.testing.assert_allclose(got_l, z["logits_samp"], atol=1e-3)
    np.testing.assert_allclose(feat.double().sum((0, 2, 3)).cpu().numpy(), z["feat_sum"], rtol=1e-3, atol=2e-2)
    iou_vs_ref = (mask == ref_mask).double().mean()
    assert float(iou_vs_ref) > 1 - 1e-4


def _sampled_grad_errors(eng, z):
    """Per-tensor L2-relative gradient error against the reference's sampled entries (up to 1024 per tensor, whole
    tensor when it has <= 1024 elements; tests/golden/gen_golden.py:gen_fcd67_train), the per-tensor norm error, and the
    whole-arena L2-relative error estimated from the same samples (each tensor's sampled squared error scaled by
    numel / samples)."""
    per, num, den = {}, 0.0, 0.0
    for m in eng.metas:
        if m.kind != 0:
            continue
        g = eng.grad_views[m.name].cpu().numpy().reshape(-1)
        idx = sample_idx(g.size, 1024, 1234).numpy()
        ref = z["gradsamp/" + m.name].astype(np.float64)
        nrm = float(z["gradnorm/" + m.name])
        d2 = float(((g[idx].astype(np.float64) - ref) ** 2).sum()) * g.size / idx.size
        # floor: a conv bias in front of BatchNorm-only consumers has an exactly-zero gradient (rounding noise both sides)
        floor = 1e-6 * np.sqrt(g.size)
        per[m.name] = (np.sqrt(d2) / max(nrm, floor), abs(float(np.linalg.norm(g)) - nrm) / max(nrm, floor))
        if nrm > floor:
            num += d2
            den += nrm * nrm
    return per, float(np.sqrt(num / den))


# (arithmetic, per-tensor bar, arena bar, median bar), set from what is measured on MI355X against this fixture (random
# labels on a random-init net: the loss sits at ln 4 and every gradient is a cancelling sum, the worst case for relative
# errors; the CPU oracle reproduces the reference's samples to 1e-6; tools/grad_ref_probe.py lists every tensor):
#   exact-fp32 kernels      median 8.1e-4, p90 2.2e-3, max 3.9e-3 (full tensors) / 4.1e-3 (1024 samples), arena 7.7e-5
#   default f16x2 / bf16x2  median 7.8e-4, p90 1.5e-3, max 2.8e-3, arena 2.6e-4
#     (before the 2^8 weight pre-scale of csrc/split16.h the low parts of the f16 weights were subnormal and this mode
#      measured median 2.0e-3 / max 1.1e-2)
# Bars = median <= 1.2e-3, every tensor <= max(5e-3, 5 x its own fp32-vs-fp64 noise floor), whole arena <= 3e-4 / 5e-4.
_TRAIN_MODES = [("fp32,fp32", 5e-3, 3e-4, 1.2e-3), (None, 5e-3, 5e-4, 1.2e-3)]


@pytest.mark.parametrize("arith,tensor_bar,arena_bar,median_bar", _TRAIN_MODES, ids=["exact_fp32", "default"])
def test_fcd67_train_steps_vs_golden(arith, tensor_bar, arena_bar, median_bar):
    """Five training steps of FCDenseNet67 (2x120x160) against the reference's own training_step + AdamW
    (fixture fcd67_train_120x160): loss / accuracy trajectory, step-0 gradients per tensor and over the whole arena,
    step-0 parameters and running statistics."""
    from sim2real_lane_segment_amd.engine import parse_dense_arith
    z = load("fcd67_train_120x160")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed, steps = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"]), int(z["steps"])
    assert steps == 5
    st = O.init_state(cfg, seed)
    eng = make_engine(cfg, st)
    if arith is not None:
        eng.set_dense_arith(*parse_dense_arith(arith))
    m_buf = torch.zeros_like(eng.params)
    v_buf = torch.zeros_like(eng.params)
    for s in range(steps):
        x, y = synth_batch(n, h, w, 4, seed + 10 * s + 1)
        if s == 0:
            y[0][y[0] == 3] = 0
        scales = O.make_drop_scales(cfg, n, seed + 10 * s + 2)
        probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
        out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
        eng.backward(1.0)
        torch.cuda.synchronize()
        assert abs(float(out[0]) - float(z["losses"][s])) < 2e-4, (s, float(out[0]), float(z["losses"][s]))
        assert abs(float(out[1]) * 100 - float(z["accs"][s])) < 0.05
        if s == 0:
            per, arena = _sampled_grad_errors(eng, z)
            l2 = np.array([v[0] for v in per.values()])
            print(f"[fcd67 train, {arith or 'default'}] gradient L2-relative error vs the reference: median "
                  f"{np.median(l2):.2e}, p90 {np.quantile(l2, 0.9):.2e}, max {l2.max():.2e}, arena {arena:.2e}")
            floors = {k: _noise_floor("fcd67_2x120x160", k) for k in per}
            bad = [(k, e, ne) for k, (e, ne) in per.items() if not (e < max(tensor_bar, 5 * floors[k]) and ne < 5e-3)]
            assert not bad, f"{len(bad)} gradient tensors off: {sorted(bad, key=lambda t: -t[1])[:10]}"
            assert arena < arena_bar, arena
            assert float(np.median(l2)) < median_bar
            grads0 = {m.name: eng.grad_views[m.name].cpu().numpy().copy() for m in eng.metas if m.kind == 0}
        eng.adamw_step(m_buf, v_buf, s + 1, 1e-3, weight_decay=1e-4)
        if s == 0:
            torch.cuda.synchronize()
            for m in eng.metas:
                if m.kind == 0:
                    p = eng.views[m.name].cpu().numpy()
                    idx = sample_idx(p.size, 64, 1234).numpy()
                    # gradsamp holds 1024 samples (same permutation: its first 64 are idx) or, for tensors of <= 1024
                    # elements, the whole tensor in order
                    gs = z["gradsamp/" + m.name]
                    gref = gs[idx] if p.size <= 1024 else gs[:idx.size]
                    got_g = grads0[m.name].reshape(-1)[idx]
                    tol = 1e-4 + np.minimum(2.1e-3, 2e-3 * np.abs(got_g - gref) / (np.abs(gref) + 1e-8))
                    assert np.all(np.abs(p.reshape(-1)[idx] - z["param1samp/" + m.name]) <= tol), m.name
            for k in z.files:
                if k.startswith("buf1/"):
                    np.testing.assert_allclose(eng.views[k[5:]].cpu().numpy(), z[k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_config0_batch8_train_step_vs_oracle():
    """BASELINE.json configs[0]: batch 8, 120x160, one full training step against the CPU oracle run live."""
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 11)
    n, h, w = 8, 120, 160
    x, y = synth_batch(n, h, w, 4, 12)
    scales = O.make_drop_scales(cfg, n, 13)
    eng = make_engine(cfg, st)
    probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
    out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, max(1, len(os.sched_getaffinity(0)))))
    ts = O.TrainState({k: v.clone() for k, v in st.items()})
    loss, acc, grads, probs_ref = O.train_step(ts, x, y, cfg, scales, apply_update=False)
    assert abs(float(out[0]) - float(loss)) < 2e-4
    assert abs(float(out[1]) * 100 - float(acc)) < 0.05
    np.testing.assert_allclose(probs.cpu().numpy(), probs_ref.numpy(), atol=1e-3)
    bad = []
    worst = 0.0
    for k, g in grads.items():
        got = eng.grad_views[k].cpu()
        floor = 1e-5 * g.numel() ** 0.5
        err = float((got - g).abs().max()) / max(float(g.abs().max()), 1e-5)
        l2 = float((got - g).norm()) / max(float(g.norm()), floor)
        nerr = abs(float(got.norm()) - float(g.norm())) / max(float(g.norm()), floor)
        worst = max(worst, l2)
        if not (err < 1e-1 and l2 < grad_l2_bar("fcd67_2x120x160", k) and nerr < 1e-2):
            bad.append((k, err, l2, nerr))
    print(f"[config0] worst L2-relative gradient error {worst:.2e}")
    assert not bad, f"{len(bad)} gradient tensors off: {bad[:10]}"


def test_two_domain_batch8_train_step_vs_oracle():
    """BASELINE.json configs[4] on one GPU: a batch mixed 50/50 from two synthetic domains by the reference's weighted
    sampler (dataManagement/dataModules.py:79-85 -> synthetic.make_two_domain_batch), one full training step against the
    CPU oracle run live.  The mixed batch has bimodal per-channel statistics: BatchNorm sees both looks at once."""
    from sim2real_lane_segment_amd.synthetic import make_two_domain_batch
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 31)
    n, h, w = 8, 120, 160
    x, y, dom = make_two_domain_batch(n, h, w, seed=32)
    assert 0 < int(dom.sum()) < n
    scales = O.make_drop_scales(cfg, n, 33)
    eng = make_engine(cfg, st)
    probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
    out, am, _ = eng.loss(probs, y.cuda(), weighted=True, want_argmax=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, max(1, len(os.sched_getaffinity(0)))))
    ts = O.TrainState({k: v.clone() for k, v in st.items()})
    loss, acc, grads, probs_ref = O.train_step(ts, x, y, cfg, scales, apply_update=False)
    assert abs(float(out[0]) - float(loss)) < 2e-4
    assert abs(float(out[1]) * 100 - float(acc)) < 0.05
    np.testing.assert_allclose(probs.cpu().numpy(), probs_ref.numpy(), atol=1e-3)
    top2 = torch.topk(probs_ref, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(am.cpu()[clear], probs_ref.argmax(1)[clear])
    errs, bad = [], []
    num = den = 0.0
    for k, g in grads.items():
        got = eng.grad_views[k].cpu()
        floor = 1e-5 * g.numel() ** 0.5
        l2 = float((got - g).norm()) / max(float(g.norm()), floor)
        errs.append(l2)
        num += float((got - g).double().pow(2).sum())
        den += float(g.double().pow(2).sum())
        if not l2 < 3e-3:  # structured frames: measured median 1.6e-4, max 1.5e-3
            bad.append((k, l2))
    print(f"[two-domain] gradient L2-relative error vs the oracle: median {np.median(errs):.2e}, max {max(errs):.2e}, "
          f"arena {np.sqrt(num / den):.2e}; domains {dom.tolist()}")
    assert not bad, f"{len(bad)} gradient tensors off: {bad[:10]}"
    assert np.sqrt(num / den) < 5e-4 and float(np.median(errs)) < 5e-4


@retry_if_not_reproducible
def test_full_size_properties_batch64():
    """BASELINE.json configs[1] size (batch 64, 120x160): size-independent properties."""
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 21)
    eng = make_engine(cfg, st)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 3, 120, 160, generator=g).cuda()
    y = torch.randint(0, 4, (64, 120, 160), generator=g).cuda()
    # eval forward is per-sample independent: a batch equals the concatenation of its halves, bit for bit
    p_all, _ = eng.forward(x, training=False)
    p_all = p_all.clone()
    p_a, _ = eng.forward(x[:32].contiguous(), training=False)
    p_a = p_a.clone()
    p_b, _ = eng.forward(x[32:].contiguous(), training=False)
    if not (torch.equal(p_all[:32], p_a) and torch.equal(p_all[32:], p_b)):
        # localise: which samples / rows / columns, and which forward changes when repeated
        msg = []
        for name, full, half in (("first", p_all[:32], p_a), ("second", p_all[32:], p_b)):
            d = (full - half).abs()
            if float(d.max()) == 0:
                continue
            bad_s = torch.nonzero(d.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
            s0 = bad_s[0]
            rows = torch.nonzero(d[s0].amax(dim=(0, 2)) > 0).flatten().tolist()
            cols = torch.nonzero(d[s0].amax(dim=(0, 1)) > 0).flatten().tolist()
            msg.append(f"{name} half: max {float(d.max()):.3e}, samples {bad_s[:40]} ({len(bad_s)}); sample {s0}: rows "
                       f"{rows[:6]}..{rows[-3:]} ({len(rows)}), cols {cols[:6]}..{cols[-3:]} ({len(cols)}), pixels "
                       f"{int((d[s0].amax(0) > 0).sum())}")
        f2 = eng.forward(x, training=False)[0].clone()
        b2 = eng.forward(x[32:].contiguous(), training=False)[0].clone()
        a2 = eng.forward(x[:32].contiguous(), training=False)[0].clone()
        msg.append(f"repeat: full==full2 {torch.equal(p_all, f2)}, b==b2 {torch.equal(p_b, b2)}, a==a2 {torch.equal(p_a, a2)}, "
                   f"full2 halves == a2/b2 {torch.equal(f2[:32], a2)}/{torch.equal(f2[32:], b2)}")
        raise AssertionError("a batch differs from the concatenation of its halves: " + " | ".join(msg))
    assert torch.allclose(p_all.sum(1), torch.ones_like(p_all[:, 0]), atol=1e-5)
    # training step twice with the same seed: bitwise reproducible loss and gradients (no float atomics)
    res = []
    for _ in range(2):
        eng.load_state(st)
        probs, _ = eng.forward(x, training=True, with_backward=True, seed=77)
        out, _, _ = eng.loss(probs, y, weighted=True)
        eng.backward(1.0)
        torch.cuda.synchronize()
        res.append((out.clone(), eng.grads.clone()))
    assert torch.equal(res[0][0], res[1][0]), f"loss outputs differ between two identical steps: {res[0][0]} vs {res[1][0]}"
    if not torch.equal(res[0][1], res[1][1]):
        off = [k for k, v in eng.grad_views.items()
               if not torch.equal(v.reshape(-1), res[0][1][v.storage_offset():v.storage_offset() + v.numel()])]
        raise AssertionError(f"gradients differ between two identical steps in {len(off)} tensors: {off[:12]}")
    assert torch.isfinite(res[0][1]).all() and float(res[0][1].abs().max()) > 0
    # gradient linearity in loss_scale
    eng.backward(2.0)
    torch.cuda.synchronize()
    assert torch.allclose(eng.grads, 2 * res[0][1], rtol=1e-5, atol=1e-12)


def test_module_api_training_step_and_optimizer():
    """The reference-shaped module: training_step -> loss.backward() -> FusedAdamW.step()."""
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import RightLaneModule, SimpleTrainModule
    assert RightLaneModule is SimpleTrainModule
    torch.manual_seed(0)
    model = SimpleTrainModule(lr=1e-3, lrRatio=1e3, decay=1e-4, num_cls=4).cuda()
    model.train()
    opt, sched = model.configure_optimizers()
    x, y = synth_batch(2, 64, 96, 4, 3)
    before = model.featureExtractor.firstconv.weight.detach().clone()
    loss = model.training_step((x.cuda(), y.cuda()), 0, seed=1)
    assert loss.requires_grad
    loss.backward()
    g = model.featureExtractor.firstconv.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    opt[0].step()
    sched[0].step()
    assert not torch.equal(before, model.featureExtractor.firstconv.weight.detach())
    sd = model.state_dict()
    assert len(sd) == 434
    model.eval()
    out = model(x.cuda())
    assert out.shape == (2, 4, 64, 96)
    ev = model.evaluate_batch((x.cuda(), y.cuda()))
    assert set(ev) == {"loss", "acc", "dice", "iou", "weight"}
    logs = model.summarize_evaluation_results([ev])
    assert 0 <= float(logs["acc"]) <= 100
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 16, 16).cuda())  # fewer than 32 px per side: five poolings do not fit


def test_mme_unlabelled_step_vs_golden():
    """MMETrainingModule.training_step(optimizer_idx=0) + the reference's SGD-nesterov groups, reduced net."""
    z = load("mme_tiny_40x56")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    xu, _ = synth_batch(n, h, w, cfg.n_classes, seed + 5)
    scales = O.make_drop_scales(cfg, n, seed + 2)
    eng = make_engine(cfg, st)
    probs, _ = eng.forward(xu.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
    out = eng.entropy_loss(probs, 0.1)
    eng.backward(1.0)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(z["loss0"])) < 1e-5
    bad = []
    for m in eng.metas:
        if m.kind == 0:
            err = rel_err(eng.grad_views[m.name].cpu().numpy(), z["grad0/" + m.name])
            if not err < 3e-3:
                bad.append((m.name, err))
    assert not bad, bad[:10]
    # SGD on the GPU's own gradients must equal the oracle's formula exactly; vs the reference within grad noise
    buf = torch.zeros_like(eng.params)
    split = [mm.offset for mm in eng.metas if mm.name == "classifier.finalConv.weight"][0]
    g_gpu = {m.name: eng.grad_views[m.name].cpu().clone() for m in eng.metas if m.kind == 0}
    p_ref = {m.name: eng.views[m.name].cpu().clone() for m in eng.metas if m.kind == 0}
    bufs = {}
    for step, key in enumerate(["param_after_sgd/", "param_after_sgd2/"]):
        eng.sgd_step(buf, 0, split, 1e-3 / 3, 0.9, 1e-4, step == 0)
        eng.sgd_step(buf, split, eng.n_param, 1e-3, 0.9, 1e-4, step == 0)
        torch.cuda.synchronize()
        for m in eng.metas:
            if m.kind != 0:
                continue
            lr = 1e-3 if m.name.startswith("classifier.") else 1e-3 / 3
            bufs[m.name] = O.sgd_nesterov_step(p_ref[m.name], g_gpu[m.name], bufs.get(m.name), lr, 0.9, 1e-4)
            np.testing.assert_allclose(eng.views[m.name].cpu().numpy(), p_ref[m.name].numpy(), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(eng.views[m.name].cpu().numpy(), z[key + m.name], rtol=1e-4, atol=2e-6,
                                       err_msg=m.name)


def test_mme_module_api():
    from sim2real_lane_segment_amd.trainingModules.MMETrainingModule import MMETrainingModule
    torch.manual_seed(0)
    model = MMETrainingModule(lr=1e-3, lrRatio=1e3, decay=1e-4, num_cls=4).cuda()
    model.train()
    (opt_g, opt_f), (sch_g, sch_f) = model.configure_optimizers()
    xl, y = synth_batch(2, 64, 96, 4, 3)
    xu, _ = synth_batch(2, 64, 96, 4, 4)
    batch = (xl.cuda(), xu.cuda(), y.cuda(), None)
    w0 = model.classifier.finalConv.weight.detach().clone()
    loss0 = model.training_step(batch, 0, optimizer_idx=0, seed=1)
    loss0.backward()
    opt_g.step()
    opt_g.zero_grad()
    loss1 = model.training_step(batch, 0, optimizer_idx=1, seed=2)
    loss1.backward()
    opt_f.step()
    assert float(loss0) < 0 < float(loss1)
    assert not torch.equal(w0, model.classifier.finalConv.weight.detach())
    assert torch.isfinite(model.featureExtractor.firstconv.weight).all()


@pytest.mark.parametrize("variant", ["57", "103"])
def test_named_variants_train_step_vs_oracle(variant):
    """FCDenseNet57 (growth 12: channel counts off the 16-channel tile grid) and FCDenseNet103 (blocks 4,5,7,10,12 /
    bottleneck 15) as the reference's factories build them (tiramisu.py:150-170): module construction through the
    package's own factory, state_dict keys/shapes against the oracle's inventory, then one training step at 2x64x96
    against the CPU oracle run live (the reference's kernels are torch's; the oracle is pinned on the 67 variant)."""
    from sim2real_lane_segment_amd.models.FCDenseNet import tiramisu as T
    down, up, bott, growth = T._VARIANTS[variant]
    cfg = O.NetConfig(down_blocks=down, up_blocks=up, bottleneck_layers=bott, growth_rate=growth, n_classes=4)
    model = (T.FCDenseNet57 if variant == "57" else T.FCDenseNet103)(4)
    spec = dict(O.state_spec(cfg))
    sd = model.state_dict()
    assert set(sd.keys()) == set(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    st = O.init_state(cfg, 21)
    n, h, w = 2, 64, 96
    x, y = synth_batch(n, h, w, 4, 22)
    scales = O.make_drop_scales(cfg, n, 23)
    eng = make_engine(cfg, st)
    torch.set_num_threads(min(16, max(1, len(os.sched_getaffinity(0)))))
    # eval forward first (initial running statistics): argmax masks equal wherever the oracle's top-2 margin is not a
    # near tie, probabilities within 1e-3
    with torch.no_grad():
        feat = O.features_forward(st, x, cfg)
        p_ref = O.classifier_forward(st, feat, cfg)
    p_gpu, _ = eng.forward(x.cuda(), training=False)
    top2 = torch.topk(p_ref, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(p_gpu.cpu().argmax(1)[clear], p_ref.argmax(1)[clear])
    np.testing.assert_allclose(p_gpu.cpu().numpy(), p_ref.numpy(), atol=1e-3)
    # one training step
    probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
    out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    ts = O.TrainState({k: v.clone() for k, v in st.items()})
    loss, acc, grads, probs_ref = O.train_step(ts, x, y, cfg, scales, apply_update=False)
    assert abs(float(out[0]) - float(loss)) < 2e-4
    np.testing.assert_allclose(probs.cpu().numpy(), probs_ref.numpy(), atol=1e-3)
    # Every tensor within 1e-2 L2-relative and the whole arena within 1e-3, whatever the tensor's own fp32-vs-fp64 noise
    # floor says (FCDenseNet103 at 64x96 has 21 tensors whose floor alone is above 1e-2: 2x3-pixel maps at the bottom of a
    # 100-layer net).  Measured on MI355X: FCDenseNet57 median 7.7e-4 / max 7.5e-3 / arena 1.9e-4, FCDenseNet103
    # median 9.4e-4 / max 4.9e-3 / arena 4.8e-4 -- the ill-conditioned tensors sit at 1.7e-3, well inside their floor.
    #
    # FCDenseNet103's gradients at this size are not continuous in the input at fp32 resolution: the oracle's OWN gradients
    # move by up to 2.7e-2 per tensor (arena 2e-3) when x is scaled by (1 + 1e-7) -- a discrete event (ReLU / max-pool
    # selection at the 2x3..8x12 levels) that any change of summation order can toggle as well (tools/pair_diff.py: a
    # build that only reorders the fp32 sums of the dense forward lands on the oracle's other branch, probabilities equal
    # to 3e-7).  So for that variant the oracle is run a second time on the scaled input and a tensor may differ by twice
    # the oracle's own sensitivity where that exceeds the fixed bound; FCDenseNet57 keeps the fixed bounds.
    sens, arena_sens = {}, 0.0
    if variant == "103":
        ts2 = O.TrainState({k: v.clone() for k, v in st.items()})
        _, _, grads2, _ = O.train_step(ts2, x * (1.0 + 1e-7), y, cfg, scales, apply_update=False)
        n2 = d2 = 0.0
        for k, g in grads.items():
            floor = 1e-5 * g.numel() ** 0.5
            sens[k] = float((grads2[k] - g).norm()) / max(float(g.norm()), floor)
            n2 += float((grads2[k] - g).double().pow(2).sum())
            d2 += float(g.double().pow(2).sum())
        arena_sens = float(np.sqrt(n2 / d2))
        print(f"[fcd103] oracle sensitivity to x*(1+1e-7): max {max(sens.values()):.2e}, arena {arena_sens:.2e}")
    table = f"fcd{variant}_2x64x96"
    bad, errs = [], []
    num = den = 0.0
    for k, g in grads.items():
        got = eng.grad_views[k].cpu()
        floor = 1e-5 * g.numel() ** 0.5
        l2 = float((got - g).norm()) / max(float(g.norm()), floor)
        errs.append(l2)
        num += float((got - g).double().pow(2).sum())
        den += float(g.double().pow(2).sum())
        if not l2 < max(1e-2, 2.0 * sens.get(k, 0.0)):
            bad.append((k, l2, sens.get(k, 0.0), _noise_floor(table, k)))
    errs = np.array(errs)
    print(f"[fcd{variant}] gradient L2-relative error vs the oracle: median {np.median(errs):.2e}, p90 "
          f"{np.quantile(errs, 0.9):.2e}, max {errs.max():.2e}, arena {np.sqrt(num / den):.2e}")
    assert not bad, f"{len(bad)} gradient tensors off: {bad[:10]}"
    assert np.sqrt(num / den) < max(1e-3, 2.0 * arena_sens) and float(np.median(errs)) < 2e-3


def test_differentiable_module_forward_matches_fused_step():
    """TrainingBase.forward in train mode carries a grad_fn: a user-written step (SimpleTrain.py:15-16 spelled out with
    torch ops on the returned probabilities) yields the same parameter gradients as the fused training_step."""
    import torch.nn.functional as F
    from sim2real_lane_segment_amd.owner import ForwardFn
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
    from sim2real_lane_segment_amd.trainingModules.TrainingBase import getClassWeight
    torch.manual_seed(1)
    model = SimpleTrainModule(num_cls=4).cuda().train()
    x, y = synth_batch(2, 64, 96, 4, 5)
    x, y = x.cuda(), y.cuda()
    loss_f = model.training_step((x, y), 0, seed=7)
    loss_f.backward()
    ref = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    probs = ForwardFn.apply(model, x, None, 7, *model._rln_params_in_arena_order())
    assert probs.requires_grad
    loss_u = F.cross_entropy(probs, y, weight=getClassWeight(y, 4).cuda())
    loss_u.backward()
    assert abs(float(loss_u) - float(loss_f)) < 1e-5
    worst = 0.0
    for n, p in model.named_parameters():
        scale = max(float(ref[n].abs().max()), 1e-6)
        worst = max(worst, float((p.grad - ref[n]).abs().max()) / scale)
    assert worst < 1e-3, worst  # measured 3.5e-4: torch differentiates softmax-of-softmax CE in fp32 on the other side
    # plain call: train mode + autograd -> differentiable; eval / no_grad -> inference tensor
    out = model(x)
    assert out.requires_grad
    with torch.no_grad():
        assert not model(x).requires_grad
    model.eval()
    assert not model(x).requires_grad


def test_stale_backward_and_standalone_pieces_fail_loudly():
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
    model = SimpleTrainModule(num_cls=4).cuda().train()
    x, y = synth_batch(1, 32, 48, 4, 9)
    x, y = x.cuda(), y.cuda()
    l1 = model.training_step((x, y), 0)
    model.training_step((x, y), 1)  # a second forward replaces the engine's activations
    with pytest.raises(RuntimeError, match="stale forward"):
        l1.backward()
    with pytest.raises(RuntimeError, match="inference-only"):
        model.featureExtractor(x)
    with torch.no_grad():
        feat = model.featureExtractor(x)
    assert feat.shape == (1, 288, 32, 48)
    with pytest.raises(RuntimeError, match="inference-only"):
        model.classifier(feat.requires_grad_(True))
    model.eval()
    assert model.classifier(feat.detach()).shape == (1, 4, 32, 48)


def test_eval_cache_reuses_tables_bit_identically():
    """rln_set_eval_cache (frozen-model loops, makeDemoVideo.py:15-47): eval forwards that reuse the weight fragments and
    folded BatchNorm tables are bit-identical to forwards that rebuild them; in-place parameter writes (torch version
    counters), engine-side optimiser steps and training forwards (new running statistics) all trigger a rebuild."""
    z = load("g16_32x48")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    eng = make_engine(cfg, O.init_state(cfg, seed))
    x, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    x = x.cuda()

    def fwd():
        out = eng.forward(x, training=False)[0].clone()
        torch.cuda.synchronize()
        return out

    p0 = fwd()
    eng.set_eval_cache(True)
    p1, p2, p3 = fwd(), fwd(), fwd()
    assert torch.equal(p0, p1) and torch.equal(p0, p2) and torch.equal(p0, p3)
    # a write through a parameter view moves the arena's version counter
    name = next(k for k in eng.views if k.endswith("conv.weight"))
    with torch.no_grad():
        eng.views[name].mul_(1.25)
    p4 = fwd()
    eng.set_eval_cache(False)
    p5 = fwd()
    assert torch.equal(p4, p5) and not torch.equal(p4, p0)
    # a training forward replaces the tables (batch statistics) and moves the running statistics
    eng.set_eval_cache(True)
    fwd()
    eng.forward(x, training=True)
    p6 = fwd()
    eng.set_eval_cache(False)
    p7 = fwd()
    assert torch.equal(p6, p7) and not torch.equal(p6, p4)
    # an engine-side optimiser step (raw-pointer write) bumps the engine's own epoch
    eng.set_eval_cache(True)
    fwd()
    eng.grads.fill_(1e-3)
    m = torch.zeros_like(eng.params)
    v = torch.zeros_like(eng.params)
    eng.adamw_step(m, v, 1, 1e-2)
    p8 = fwd()
    eng.set_eval_cache(False)
    p9 = fwd()
    assert torch.equal(p8, p9) and not torch.equal(p8, p6)


@retry_if_not_reproducible
def test_workspace_recarve_is_ordered_after_inflight_kernels():
    """A forward with a new geometry re-carves the workspace, usually on the memory the previous geometry's kernels are still
    using when the host runs ahead (a cold GPU, or any device backlog).  rln_set_workspace used to write its descriptor
    tables into that block with host copies that HIP does not order after the stream: 1 in ~15 fresh-box runs of the batch-64
    property test failed.  Here a few hundred ms of device work are queued in front of a batch forward that is followed at
    once by the forward of its second half (tools/ws_race_probe.py: the unfixed library fails this within six rounds)."""
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 21)
    eng = make_engine(cfg, st)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 3, 120, 160, generator=g).cuda()
    ref_all = eng.forward(x, training=False)[0].clone()
    ref_b = eng.forward(x[32:].contiguous(), training=False)[0].clone()
    assert torch.equal(ref_all[32:], ref_b)
    big = torch.randn(12288, 12288, device="cuda")
    for it in range(6):
        eng.forward(x[:2].contiguous(), training=False)  # another geometry: the batch forward below re-carves too
        torch.cuda.synchronize()
        for _ in range(24):  # ~0.6 s of device backlog in front of the forward
            big @ big
        p_all = eng.forward(x, training=False)[0]
        p_b = eng.forward(x[32:].contiguous(), training=False)[0]
        torch.cuda.synchronize()
        assert torch.equal(p_all, ref_all), f"round {it}: the batch forward changed under a device backlog"
        if not torch.equal(p_b, ref_b):
            d = (p_b - ref_b).abs()
            bad_s = torch.nonzero(d.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
            s0 = bad_s[0]
            rows = torch.nonzero(d[s0].amax(dim=(0, 2)) > 0).flatten().tolist()
            cols = torch.nonzero(d[s0].amax(dim=(0, 1)) > 0).flatten().tolist()
            again = eng.forward(x[32:].contiguous(), training=False)[0].clone()       # same workspace
            eng.forward(x[:2].contiguous(), training=False)
            again2 = eng.forward(x[32:].contiguous(), training=False)[0].clone()      # re-carved on a quiet device
            raise AssertionError(
                f"round {it}: the half-batch forward after a re-carve changed: max {float(d.max()):.3e}, samples {bad_s[:20]} "
                f"({len(bad_s)}); sample {s0}: rows {rows[:4]}..{rows[-2:]} ({len(rows)}), cols {cols[:4]}..{cols[-2:]} ({len(cols)}), "
                f"pixels {int((d[s0].amax(0) > 0).sum())}; repeated on the same workspace == reference {torch.equal(again, ref_b)}, "
                f"== the bad one {torch.equal(again, p_b)}; re-carved quietly == reference {torch.equal(again2, ref_b)}")
