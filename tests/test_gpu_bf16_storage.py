"""bf16 storage mode (-m gpu): activation stacks and finalised output gradients of the levels that carry the traffic are
bf16 planes in HBM (rln_set_storage(ctx, 1); BASELINE.json configs[1] "bf16", configs[3] "fp16": the reference reaches
mixed precision through Lightning's --precision 16, train.py:100-101).  Operands are plain bf16, accumulation / BatchNorm
statistics / gradient stacks / parameters stay fp32.

This is NOT the parity mode: the fp32-storage default keeps the bit-exact-mask / 1e-3 bar (test_gpu_parity.py).  Here the
tests report how far 8-bit mantissas move the result from the reference's fp32 numbers and assert the measured level
(SURVEY.md section 7 measured the same effect for CPU bf16 autocast of the reference: 0.26 % of the pixels flip, mask IoU
0.9941 on random-init weights)."""
import os

import numpy as np
import pytest
import torch

from oracle import fcdensenet_oracle as O
from tests.golden.common import cfg_from_arrays, retry_if_not_reproducible, synth_batch, unpack_masks

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def make_engine(cfg, st, storage):
    from sim2real_lane_segment_amd.engine import Engine, NetSpec
    spec = NetSpec(in_channels=cfg.in_channels, down_blocks=cfg.down_blocks, up_blocks=cfg.up_blocks,
                   bottleneck_layers=cfg.bottleneck_layers, growth_rate=cfg.growth_rate,
                   out_chans_first_conv=cfg.out_chans_first_conv, n_classes=cfg.n_classes,
                   temperature=cfg.temperature)
    eng = Engine(spec, device="cuda")
    if storage == "bf16":
        eng.set_storage("bf16")
    elif storage == "f32_bf16x1":  # same one-part bf16 operands, fp32 stacks: isolates the storage rounding
        eng.set_dense_arith(1, "bf16", 1, "bf16")
    eng.load_state(st)
    return eng


@pytest.mark.parametrize("name", ["fcd67_eval_120x160", "fcd67_eval_480x640"])
def test_bf16_storage_eval_masks_vs_reference(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, _ = synth_batch(n, h, w, 4, seed + 1)
    ref_mask = unpack_masks(z["mask_packed"], n * h * w)
    idx = torch.from_numpy(z["sample_idx"])
    res = {}
    for storage in ("f32_bf16x1", "bf16"):
        eng = make_engine(cfg, st, storage)
        probs, _ = eng.forward(x.cuda(), training=False)
        torch.cuda.synchronize()
        assert torch.isfinite(probs).all()
        assert torch.allclose(probs.sum(1), torch.ones_like(probs[:, 0]), atol=1e-5)
        mask = probs.argmax(1).reshape(-1).cpu()
        agree = float((mask == ref_mask).double().mean())
        perr = float(np.abs(probs.permute(0, 2, 3, 1).reshape(-1, 4).cpu()[idx].numpy() - z["probs_samp"]).max())
        res[storage] = (agree, perr)
    print(f"[{name}] mask agreement with the reference / max |dp| on 1024 sampled pixels: fp32 stacks + bf16 operands "
          f"{res['f32_bf16x1'][0]:.5f} / {res['f32_bf16x1'][1]:.3f}; bf16 stacks {res['bf16'][0]:.5f} / {res['bf16'][1]:.3f}")
    # measured on MI355X (random-init weights, |scaled logit| gaps of 1e-2 on 2.8 % of the pixels): see DESIGN.md 4.3
    assert res["bf16"][0] > 0.98
    assert res["bf16"][1] < 0.25


def test_bf16_storage_train_step_vs_oracle():
    """One training step at 2x120x160 against the CPU oracle: loss, probabilities, per-tensor / arena gradient error."""
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 41)
    n, h, w = 2, 120, 160
    x, y = synth_batch(n, h, w, 4, 42)
    scales = O.make_drop_scales(cfg, n, 43)
    torch.set_num_threads(min(16, max(1, len(os.sched_getaffinity(0)))))
    ts = O.TrainState({k: v.clone() for k, v in st.items()})
    loss, acc, grads, probs_ref = O.train_step(ts, x, y, cfg, scales, apply_update=False)
    out_line = []
    for storage in ("f32_bf16x1", "bf16"):
        eng = make_engine(cfg, st, storage)
        probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
        out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
        eng.backward(1.0)
        torch.cuda.synchronize()
        assert torch.isfinite(eng.grads).all()
        errs = []
        num = den = 0.0
        for k, g in grads.items():
            got = eng.grad_views[k].cpu()
            floor = 1e-5 * g.numel() ** 0.5
            errs.append(float((got - g).norm()) / max(float(g.norm()), floor))
            num += float((got - g).double().pow(2).sum())
            den += float(g.double().pow(2).sum())
        arena = float(np.sqrt(num / den))
        dl = abs(float(out[0]) - float(loss))
        dp = float((probs.cpu() - probs_ref).abs().max())
        out_line.append(f"{storage}: |dloss| {dl:.2e}, max|dp| {dp:.2e}, grad L2 median {np.median(errs):.2e} "
                        f"max {max(errs):.2e} arena {arena:.2e}")
        if storage == "bf16":
            assert dl < 2e-2 and dp < 0.3
            assert arena < 0.25 and np.median(errs) < 0.25
    print("[bf16 storage, train step vs oracle] " + " | ".join(out_line))


@retry_if_not_reproducible
def test_bf16_storage_properties_batch64():
    """BASELINE.json configs[1] size: determinism, batch independence in eval, loss-scale linearity, finite gradients."""
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 21)
    eng = make_engine(cfg, st, "bf16")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 3, 120, 160, generator=g).cuda()
    y = torch.randint(0, 4, (64, 120, 160), generator=g).cuda()
    p_all, _ = eng.forward(x, training=False)
    p_all = p_all.clone()
    p_a, _ = eng.forward(x[:32].contiguous(), training=False)
    p_a = p_a.clone()
    p_b, _ = eng.forward(x[32:].contiguous(), training=False)
    assert torch.equal(p_all[:32], p_a) and torch.equal(p_all[32:], p_b)
    res = []
    for _ in range(2):
        eng.load_state(st)
        probs, _ = eng.forward(x, training=True, with_backward=True, seed=77)
        out, _, _ = eng.loss(probs, y, weighted=True)
        eng.backward(1.0)
        torch.cuda.synchronize()
        res.append((out.clone(), eng.grads.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.isfinite(res[0][1]).all() and float(res[0][1].abs().max()) > 0
    # loss-scale linearity: a power of two commutes with every rounding (bf16 stores, bf16 operands, fp32 sums)
    eng.backward(2.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.grads, 2 * res[0][1])


def test_bf16_storage_trains():
    """Ten AdamW steps on a fixed structured batch: the loss goes down as it does with fp32 stacks."""
    from sim2real_lane_segment_amd.synthetic import make_batch
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 3)
    x, y = make_batch(8, seed=9)
    traj = {}
    for storage in ("f32", "bf16"):
        eng = make_engine(cfg, st, storage)
        m = torch.zeros_like(eng.params)
        v = torch.zeros_like(eng.params)
        losses = []
        for s in range(10):
            probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, seed=100 + s)
            out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
            eng.backward(1.0)
            eng.adamw_step(m, v, s + 1, 1e-3, weight_decay=1e-4)
            losses.append(float(out[0]))
        traj[storage] = losses
    print(f"[bf16 storage] loss over 10 steps: fp32 stacks {traj['f32'][0]:.4f} -> {traj['f32'][-1]:.4f}, "
          f"bf16 stacks {traj['bf16'][0]:.4f} -> {traj['bf16'][-1]:.4f}")
    assert traj["bf16"][-1] < traj["bf16"][0] - 0.05
    assert abs(traj["bf16"][-1] - traj["f32"][-1]) < 0.05


@pytest.mark.parametrize("n,h,w", [(2, 240, 320), (3, 232, 312), (1, 96, 128), (2, 128, 160), (2, 88, 136)])
def test_bf16_storage_other_geometries(n, h, w):
    """Frames other than 120x160 / 480x640: levels whose width is neither 40 nor a multiple of 80 stay fp32 (the data
    gradient's pull kernel covers those widths only: 232x312, 96x128 and 88x136 keep every stack in fp32; 240x320 and
    128x160 mix), ragged tiles, odd deep levels.  bf16 stacks against the same one-part bf16 operands on fp32 stacks (the difference is
    the storage rounding alone) and against the exact-fp32 kernel family; eval masks, training loss, gradient arena.  The
    paired dense forward and the 8-pixel-unit weight gradient run in both one-part modes, at tile edges the bench size
    never produces."""
    from sim2real_lane_segment_amd.engine import Engine, NetSpec, parse_dense_arith
    cfg = O.NetConfig()
    st = O.init_state(cfg, 3)
    g = torch.Generator().manual_seed(5 + h)
    x = torch.randn(n, 3, h, w, generator=g).cuda()
    y = torch.randint(0, 4, (n, h, w), generator=g).cuda()
    scales = O.make_drop_scales(cfg, n, 11)
    res = {}
    for mode in ("exact", "f32_bf16x1", "bf16"):
        if mode == "exact":
            eng = Engine(NetSpec(n_classes=4), device="cuda", dense_arith=parse_dense_arith("fp32,fp32"))
            eng.load_state(st)
        else:
            eng = make_engine(cfg, st, mode)
        probs, _ = eng.forward(x, training=False)
        p_eval = probs.float().cpu()
        probs_t, _ = eng.forward(x, training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
        out, _, _ = eng.loss(probs_t, y, weighted=True)
        eng.backward(1.0)
        torch.cuda.synchronize()
        assert torch.isfinite(eng.grads).all()
        res[mode] = (p_eval, float(out[0]), eng.grads.clone().cpu())
    pe, le, ge = res["exact"]
    line = []
    for mode in ("f32_bf16x1", "bf16"):
        p, l, gr = res[mode]
        agree = float((p.argmax(1) == pe.argmax(1)).float().mean())
        dp = float((p - pe).abs().max())
        ga = float((gr - ge).norm() / ge.norm())
        line.append(f"{mode}: masks {agree:.4f}, max|dp| {dp:.3f}, |dloss| {abs(l - le):.1e}, grad arena {ga:.2e}")
        # measured on MI355X: masks 0.9985 .. 0.9995, max|dp| <= 0.003, |dloss| <= 2e-5, arena 1.0e-2 .. 6.1e-2
        assert agree > 0.995 and dp < 0.02 and abs(l - le) < 1e-3 and ga < 0.15
    print(f"[bf16 storage {n}x{h}x{w}] " + " | ".join(line))
