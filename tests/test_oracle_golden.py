"""Pins the CPU oracle (oracle/fcdensenet_oracle.py) against golden vectors generated from the
reference's own model / training_step code (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import fcdensenet_oracle as O
from tests.golden.common import synth_batch, sample_idx, unpack_masks, cfg_from_arrays

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"fixture {name} not generated")
    return np.load(path)


def test_state_spec_fcd67():
    cfg = O.fcdensenet67_config(4)
    spec = O.state_spec(cfg)
    assert len(spec) == 434
    n_params = sum(int(np.prod(s)) for k, s in spec if O.is_param(k))
    assert n_params == 3461220
    assert sum(1 for k, _ in spec if O.is_param(k)) == 254
    assert O.feature_channels(cfg) == 288
    assert len(O.dropout_channels(cfg)) == 60


def test_misc_adamw_cosine_classweight():
    z = load("misc")
    p = torch.from_numpy(z["adamw_p0"].copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        O.adamw_step(p, torch.from_numpy(z["adamw_grads"][s]), m, v, s + 1, 1e-3, weight_decay=1e-4)
        np.testing.assert_allclose(p.numpy(), z[f"adamw_p{s + 1}"], rtol=1e-6, atol=1e-7)
    lrs = np.array([O.cosine_lr(e, 1e-3, 1e3) for e in range(51)])
    np.testing.assert_allclose(lrs, z["cosine_lr"], rtol=1e-9, atol=1e-12)
    w = O.get_class_weight(torch.from_numpy(z["cw_targets"]), 4)
    np.testing.assert_array_equal(w.numpy(), z["cw_weights"])
    assert torch.isinf(w[3])
    with pytest.raises(AssertionError):
        O.get_class_weight(torch.tensor([0, 1, 4]), 4)


@pytest.mark.parametrize("name,full", [("tiny_40x56", True), ("tiny_33x47", True), ("g16_32x48", False),
                                       ("g16_absent_30x34", False)])
def test_small_nets(name, full):
    z = load(name)
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    if int(z["absent_class"]):
        y[y == cfg.n_classes - 1] = 0
    scales = O.make_drop_scales(cfg, n, seed + 2)

    with torch.no_grad():
        feat = O.features_forward(st, x, cfg)
        probs = O.classifier_forward(st, feat, cfg)
        logits = O.classifier_forward(st, feat, cfg, use_softmax=False)
    np.testing.assert_allclose(probs.numpy(), z["eval_probs"], atol=2e-6)
    np.testing.assert_allclose(logits.numpy(), z["eval_logits"], atol=2e-5)
    np.testing.assert_allclose(feat.sum((0, 2, 3)).numpy(), z["eval_feat_sum"], rtol=1e-4, atol=1e-4)

    ts = O.TrainState({k: v.clone() for k, v in st.items()})
    loss, acc, grads, probs_t = O.train_step(ts, x, y, cfg, scales, lr=1e-3, weight_decay=1e-4)
    assert abs(float(loss) - float(z["train_loss"])) < 1e-5
    assert abs(float(acc) - float(z["train_acc"])) < 1e-4
    ref_pt = z["train_probs"]
    got_pt = probs_t.numpy() if full else probs_t.numpy()[:, :, ::7, ::5]
    np.testing.assert_allclose(got_pt, ref_pt, atol=2e-6)
    for k, g in grads.items():
        if full:
            ref = z["grad/" + k]
            np.testing.assert_allclose(g.numpy(), ref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(ref).max())
            np.testing.assert_allclose(ts.st[k].numpy(), z["param1/" + k], rtol=1e-4, atol=2e-5)
        else:
            idx = sample_idx(g.numel(), 64, 1234)
            ref = z["gradsamp/" + k]
            np.testing.assert_allclose(g.reshape(-1)[idx].numpy(), ref, rtol=1e-3,
                                       atol=1e-6 + 1e-4 * float(z["gradnorm/" + k]))
            assert abs(float(g.norm()) - float(z["gradnorm/" + k])) <= 1e-4 * float(z["gradnorm/" + k]) + 1e-9
    for k in z.files:
        if k.startswith("buf1/"):
            np.testing.assert_allclose(ts.st[k[5:]].numpy(), z[k], rtol=1e-5, atol=1e-6)


def test_fcd67_eval_masks():
    z = load("fcd67_eval_120x160")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    x, _ = synth_batch(n, h, w, 4, seed + 1)
    with torch.no_grad():
        feat = O.features_forward(st, x, cfg)
        logits = O.classifier_forward(st, feat, cfg, use_softmax=False)
        probs = O.classifier_forward(st, feat, cfg)
    mask = torch.max(probs, 1)[1].reshape(-1)
    ref_mask = unpack_masks(z["mask_packed"], n * h * w)
    near = set(int(i) for i in z["near_tie_idx"])
    diff = torch.nonzero(mask != ref_mask).reshape(-1).tolist()
    assert all(d in near for d in diff), f"{len(diff)} mask flips outside the listed near ties"
    idx = torch.from_numpy(z["sample_idx"])
    got = probs.permute(0, 2, 3, 1).reshape(-1, 4)[idx].numpy()
    np.testing.assert_allclose(got, z["probs_samp"], atol=1e-5)
    got_l = logits.permute(0, 2, 3, 1).reshape(-1, 4)[idx].numpy()
    np.testing.assert_allclose(got_l, z["logits_samp"], atol=1e-4)


def test_mme_unlabelled_step_and_sgd():
    z = load("mme_tiny_40x56")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    st = O.init_state(cfg, seed)
    xu, _ = synth_batch(n, h, w, cfg.n_classes, seed + 5)
    scales = O.make_drop_scales(cfg, n, seed + 2)
    loss, grads = O.mme_unlabelled_step(st, xu, cfg, scales, 0.1)
    assert abs(float(loss) - float(z["loss0"])) < 1e-6
    for k, g in grads.items():
        ref = z["grad0/" + k]
        np.testing.assert_allclose(g.numpy(), ref, rtol=1e-3, atol=1e-7 + 1e-4 * np.abs(ref).max(), err_msg=k)
    # two SGD-nesterov steps with the reference's two lr groups (features lr/3, classifier lr)
    bufs = {}
    for step, key in enumerate(["param_after_sgd/", "param_after_sgd2/"]):
        for k in grads:
            lr = 1e-3 if k.startswith("classifier.") else 1e-3 / 3
            bufs[k] = O.sgd_nesterov_step(st[k], torch.from_numpy(z["grad0/" + k]), bufs.get(k), lr, 0.9, 1e-4)
            np.testing.assert_allclose(st[k].numpy(), z[key + k], rtol=1e-5, atol=1e-7, err_msg=k)


def test_fcd67_five_training_steps_vs_reference_fixture():
    """The oracle's training step against the reference's own SimpleTrainModule.training_step + AdamW over FIVE steps of
    FCDenseNet67 (fixture fcd67_train_120x160, generated by tests/golden/gen_golden.py from the reference's classes):
    loss / accuracy trajectory, step-0 gradients (norm and up to 1024 sampled entries per tensor), step-0 parameters and
    running statistics."""
    z = load("fcd67_train_120x160")
    cfg = cfg_from_arrays(z, O.NetConfig)
    n, h, w, seed, steps = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"]), int(z["steps"])
    assert steps == 5
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ts = O.TrainState(O.init_state(cfg, seed))
    for s in range(steps):
        x, y = synth_batch(n, h, w, 4, seed + 10 * s + 1)
        if s == 0:
            y[0][y[0] == 3] = 0
        scales = O.make_drop_scales(cfg, n, seed + 10 * s + 2)
        loss, acc, grads, _ = O.train_step(ts, x, y, cfg, scales, lr=1e-3, weight_decay=1e-4)
        assert abs(float(loss) - float(z["losses"][s])) < 2e-5, (s, float(loss), float(z["losses"][s]))
        assert abs(float(acc) - float(z["accs"][s])) < 2e-3
        if s == 0:
            num = den = 0.0
            for k, g in grads.items():
                idx = sample_idx(g.numel(), 1024, 1234)
                ref = z["gradsamp/" + k]
                nrm = float(z["gradnorm/" + k])
                got = g.reshape(-1)[idx].numpy()
                d2 = float(((got.astype(np.float64) - ref) ** 2).sum()) * g.numel() / len(idx)
                num += d2
                den += nrm * nrm
                # same algorithm, same operators, possibly another thread count: fp32 summation-order noise only
                assert np.sqrt(d2) <= 3e-3 * nrm + 1e-6 * np.sqrt(g.numel()), k
                assert abs(float(g.norm()) - nrm) <= 1e-3 * nrm + 1e-9, k
            assert np.sqrt(num / den) < 2e-4
            for k in z.files:
                if k.startswith("buf1/"):
                    np.testing.assert_allclose(ts.st[k[5:]].numpy(), z[k], rtol=1e-5, atol=1e-6)
