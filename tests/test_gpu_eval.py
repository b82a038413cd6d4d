"""Evaluation step and demo path on the GPU against the CPU oracle.

evaluate_batch (TrainingBase.py:79-96): eval forward, UNWEIGHTED cross_entropy on the probabilities, argmax,
accuracy / dice_score / iou each times the batch size; the device confusion matrix behind the three metrics; the
epoch summary (TrainingBase.py:98-110).  The metric functions themselves are pytorch_lightning 1.2.1's (absent
everywhere): oracle and product restate the same published definitions -> parity of those definitions is unpinned,
parity of the numbers they are applied to (loss, argmax, confusion counts) is what these tests pin."""
import numpy as np
import pytest
import torch

from oracle import fcdensenet_oracle as O
from oracle import transforms_oracle as TO
from tests.golden.common import cfg_from_arrays, synth_batch
from tests.test_gpu_parity import load, make_engine

pytestmark = pytest.mark.gpu


def _eval_case(cfg, st, x, y, tol_loss=2e-5):
    """returns (device outputs, oracle outputs) of one evaluate_batch on identical weights and inputs"""
    from sim2real_lane_segment_amd import metrics
    eng = make_engine(cfg, st)
    probs, _ = eng.forward(x.cuda(), training=False)
    out, am, conf = eng.loss(probs, y.cuda(), weighted=False, want_argmax=True, want_confusion=True)
    torch.cuda.synchronize()
    ref = O.evaluate_batch(st, x, y, cfg)
    ref_probs = O.forward(st, x, cfg, training=False)
    ref_pred = torch.max(ref_probs, 1)[1]
    ref_conf = O.confusion(ref_pred, y, cfg.n_classes)
    w = x.shape[0]
    # unweighted CE on the probabilities
    assert abs(float(out[0]) * w - float(ref["loss"])) <= tol_loss * w, (float(out[0]) * w, float(ref["loss"]))
    # argmax: report the actual flip count against the oracle (fp32 summation order differs); expect 0 on these nets
    flips = int((am.cpu() != ref_pred).sum())
    gap = torch.topk(ref_probs, 2, dim=1).values
    near = int(((gap[:, 0] - gap[:, 1]) < 1e-5).sum())
    print(f"[eval] argmax flips vs oracle: {flips} (pixels with top-2 probability gap < 1e-5: {near})")
    assert flips <= near
    if flips == 0:
        assert torch.equal(conf.cpu(), ref_conf)
    else:
        assert int((conf.cpu() - ref_conf).abs().sum()) <= 2 * flips
    assert int(conf.sum()) == y.numel() and torch.equal(conf.sum(1).cpu(), torch.bincount(y.reshape(-1), minlength=cfg.n_classes))
    got = {"acc": metrics.accuracy_from_confusion(conf) * w, "dice": metrics.dice_from_confusion(conf) * w,
           "iou": metrics.iou_from_confusion(conf) * w}
    slack = 4.0 * flips / y.numel() * w + 1e-6 * w
    for k in ("acc", "dice", "iou"):
        assert abs(float(got[k]) - float(ref[k])) <= slack + 1e-6, (k, float(got[k]), float(ref[k]))
    assert abs(float(out[1]) - float(ref["acc"]) / w) <= slack + 1e-6  # accuracy straight from the loss kernel
    return got, ref


@pytest.mark.parametrize("case", ["all_classes", "absent_in_target", "absent_in_both", "single_class_target"])
def test_evaluate_batch_small_net_vs_oracle(case):
    z = load("g16_32x48")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    k = cfg.n_classes - 1
    if case in ("absent_in_target", "absent_in_both"):
        y[y == k] = 0
    if case == "absent_in_both":  # class k is never predicted either: its logit is pushed far down
        st = dict(st)
        st["classifier.finalConv.bias"] = st["classifier.finalConv.bias"].clone()
        st["classifier.finalConv.bias"][k] = -100.0
    if case == "single_class_target":
        y[:] = 1
    got, ref = _eval_case(cfg, st, x, y)
    if case == "absent_in_both":
        # iou: num_classes is inferred as max(pred, target) + 1 -> the mean runs over k classes only
        assert float(ref["iou"]) > 0


def test_evaluate_batch_module_fcd67_vs_oracle():
    """TrainingBase.evaluate_batch / summarize_evaluation_results through the module API, FCDenseNet67 at 120x160."""
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
    z = load("fcd67_eval_120x160")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, y = synth_batch(n, h, w, 4, seed + 1)
    y[1][y[1] == 3] = 2  # second sample has no obstacle pixels
    model = SimpleTrainModule(num_cls=4)
    model.load_state_dict(st)
    model = model.cuda().eval()
    ev = model.evaluate_batch((x.cuda(), y.cuda()))
    ev2 = model.evaluate_batch((x[:1].cuda(), y[:1].cuda()))
    torch.cuda.synchronize()
    ref = O.evaluate_batch(st, x, y, cfg)
    ref2 = O.evaluate_batch(st, x[:1], y[:1], cfg)
    near = len(z["near_tie_idx"])
    slack = 4.0 * near / y.numel() * n
    assert ev["weight"] == ref["weight"] == n
    assert abs(float(ev["loss"]) - float(ref["loss"])) < 1e-4 * n
    for k in ("acc", "dice", "iou"):
        assert abs(float(ev[k]) - float(ref[k])) <= slack, (k, float(ev[k]), float(ref[k]))
    logs = model.summarize_evaluation_results([ev, ev2])          # TrainingBase.py:98-110
    tw = ref["weight"] + ref2["weight"]
    assert abs(float(logs["loss"]) - float((ref["loss"] + ref2["loss"]) / tw)) < 1e-4
    assert abs(float(logs["acc"]) - float((ref["acc"] + ref2["acc"]) / tw * 100)) <= 100 * slack
    assert abs(float(logs["iou"]) - float((ref["iou"] + ref2["iou"]) / tw * 100)) <= 100 * slack
    assert abs(float(logs["dice"]) - float((ref["dice"] + ref2["dice"]) / tw)) <= slack


def test_bad_labels_are_counted():
    """labels >= num_cls: the reference asserts in getClassWeight (TrainingBase.py:15); here out[2] counts them."""
    z = load("g16_32x48")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    y[0, 0, :7] = cfg.n_classes + 2
    eng = make_engine(cfg, st)
    probs, _ = eng.forward(x.cuda(), training=False)
    out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
    assert float(out[2]) == 7.0


def test_overlay_vs_numpy_oracle():
    """makeDemoVideo.py:36-46: argmax + resized frame + class colours, uint8 HWC."""
    from sim2real_lane_segment_amd.demo import overlay
    rng = np.random.default_rng(5)
    n, hs, ws, h, w = 3, 480, 640, 120, 160
    frames = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    logits = rng.normal(size=(n, 4, h, w)).astype(np.float32)
    logits[0, :, :5, :] = 0.25  # exact ties: the first maximum (class 0) wins, as torch.max does
    probs = torch.softmax(torch.from_numpy(logits), 1)
    out, pred = overlay(torch.from_numpy(frames).cuda(), probs.cuda(), want_pred=True)
    torch.cuda.synchronize()
    for i in range(n):
        ref_out, ref_pred = TO.overlay(frames[i], probs[i].numpy())
        assert np.array_equal(pred[i].cpu().numpy(), ref_pred)
        assert np.array_equal(pred[i].cpu().numpy(), torch.max(probs[i:i + 1], 1)[1][0].numpy().astype(np.uint8))
        assert np.array_equal(out[i].cpu().numpy(), ref_out)
    # odd source size / no resize
    f2 = rng.integers(0, 256, (1, 120, 160, 3), dtype=np.uint8)
    o2 = overlay(torch.from_numpy(f2).cuda(), probs[:1].cuda())
    assert np.array_equal(o2[0].cpu().numpy(), TO.overlay(f2[0], probs[0].numpy())[0])


def test_predict_frames_demo_path():
    """transform -> model.forward -> argmax -> painted frame, all on device, vs the oracle chain."""
    from sim2real_lane_segment_amd.demo import predict_frames
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 3)
    model = SimpleTrainModule(num_cls=4)
    model.load_state_dict(st)
    model = model.cuda().eval()
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)
    out, pred = predict_frames(model, torch.from_numpy(frames).cuda(), want_pred=True)
    torch.cuda.synchronize()
    xs = torch.from_numpy(np.stack([TO.transform(f)[0] for f in frames]))
    ref_probs = O.forward(st, xs, cfg, training=False)
    top2 = torch.topk(ref_probs, 2, dim=1).values
    sure = ((top2[:, 0] - top2[:, 1]) > 1e-4).numpy()
    ref_pred = ref_probs.argmax(1).numpy().astype(np.uint8)
    got = pred.cpu().numpy()
    assert np.array_equal(got[sure], ref_pred[sure])
    print(f"[demo] argmax flips vs oracle: {int((got != ref_pred).sum())} of {got.size}")
    for i in range(2):
        ref_out, _ = TO.overlay(frames[i], torch.nn.functional.one_hot(torch.from_numpy(got[i].astype(np.int64)), 4)
                                .permute(2, 0, 1).float().numpy())
        assert np.array_equal(out[i].cpu().numpy(), ref_out)
