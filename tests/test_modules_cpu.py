"""CPU-only checks of the reference-shaped Python surface."""
import argparse
import io

import numpy as np
import pytest
import torch

from oracle import fcdensenet_oracle as O
from sim2real_lane_segment_amd import install_aliases, metrics
from sim2real_lane_segment_amd.models.FCDenseNet import tiramisu
from sim2real_lane_segment_amd.trainingModules.MMETrainingModule import MMETrainingModule, adentropy
from sim2real_lane_segment_amd.trainingModules.SimpleTrain import RightLaneModule, SimpleTrainModule
from sim2real_lane_segment_amd.trainingModules.TrainingBase import TrainingBase, getClassWeight


def test_state_dict_keys_shapes_and_roundtrip():
    m = SimpleTrainModule(lr=2e-3, decay=1e-5, lrRatio=100, num_cls=4)
    assert RightLaneModule is SimpleTrainModule
    sd = m.state_dict()
    spec = O.state_spec(O.fcdensenet67_config(4))
    assert list(sd.keys()) == [k for k, _ in spec]
    assert all(tuple(sd[k].shape) == tuple(s) for k, s in spec)
    assert dict(m.hparams) == {"lr": 2e-3, "decay": 1e-5, "lrRatio": 100}
    assert (m.lr, m.decay, m.lrRatio, m.num_cls) == (2e-3, 1e-5, 100, 4)
    assert m.featureExtractor.getFeatureChannels() == 288
    # parameters become views of one flat arena, values preserved, Parameter identity preserved
    p0 = m.featureExtractor.firstconv.weight
    before = {k: v.clone() for k, v in sd.items()}
    eng = m._rln_sync()
    assert m.featureExtractor.firstconv.weight is p0
    assert p0.data_ptr() == eng.params.data_ptr()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    # load_state_dict writes through to the arena
    st = O.init_state(O.fcdensenet67_config(4), 3)
    m.load_state_dict(st)
    assert torch.equal(eng.views["classifier.finalConv.weight"], st["classifier.finalConv.weight"])
    assert torch.equal(eng.params[:48 * 27].view(48, 3, 3, 3), st["featureExtractor.firstconv.weight"])
    # torch.save / load of the state_dict (train.py:74-75, train.py:58)
    buf = io.BytesIO()
    torch.save(m.state_dict(), buf)
    buf.seek(0)
    m2 = MMETrainingModule(num_cls=4)
    m2.load_state_dict(torch.load(buf))
    assert torch.equal(m2.classifier.finalConv.bias, m.classifier.finalConv.bias)


def test_checkpoint_loading_like_test_py(tmp_path):
    m = SimpleTrainModule(num_cls=4)
    path = tmp_path / "model.ckpt"
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": {"lr": 5e-4, "decay": 1e-4, "lrRatio": 10.0}}, path)
    m2 = SimpleTrainModule.load_from_checkpoint(checkpoint_path=str(path), num_cls=4)  # test.py:30
    assert m2.num_cls == 4 and m2.lr == 5e-4
    assert torch.equal(m2.featureExtractor.firstconv.weight, m.featureExtractor.firstconv.weight)


def test_checkpoint_with_namespace_hparams_and_hostile_pickle(tmp_path):
    """weights-only loader: an argparse.Namespace of numbers is admitted, a pickled callable is rejected loudly."""
    import pickle
    m = SimpleTrainModule(num_cls=4)
    ok = tmp_path / "ns.ckpt"
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": argparse.Namespace(lr=2e-3, decay=1e-4, lrRatio=5.0)},
               ok)
    m2 = SimpleTrainModule.load_from_checkpoint(checkpoint_path=str(ok), num_cls=4)
    assert m2.lr == 2e-3 and m2.lrRatio == 5.0

    marker = tmp_path / "executed"

    class Hostile:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    bad = tmp_path / "bad.ckpt"
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": {"lr": Hostile()}}, bad)
    with pytest.raises(RuntimeError, match="weights-only"):
        SimpleTrainModule.load_from_checkpoint(checkpoint_path=str(bad), num_cls=4)
    assert not marker.exists()
    raw = tmp_path / "raw.ckpt"
    raw.write_bytes(pickle.dumps({"state_dict": {}, "hyper_parameters": Hostile()}))
    with pytest.raises(RuntimeError, match="weights-only"):
        SimpleTrainModule.load_from_checkpoint(checkpoint_path=str(raw), num_cls=4)
    assert not marker.exists()


def test_cli_flags_and_optimizers():
    parser = TrainingBase.add_model_specific_args(argparse.ArgumentParser())
    a = parser.parse_args(["-lr", "0.01", "--decay", "0.001", "--lrRatio", "50"])
    assert (a.learningRate, a.decay, a.lrRatio) == (0.01, 0.001, 50)
    d = parser.parse_args([])
    assert (d.learningRate, d.decay, d.lrRatio) == (1e-3, 1e-4, 1000)
    m = SimpleTrainModule(lr=1e-3, decay=1e-4, lrRatio=1e3, num_cls=4)
    (opt,), (sched,) = m.configure_optimizers()
    assert opt.param_groups[0]["weight_decay"] == 1e-4 and opt.param_groups[0]["betas"] == (0.9, 0.999)
    assert sum(p.numel() for p in opt.param_groups[0]["params"]) == 3461220
    z = np.load("tests/golden/misc.npz")
    lrs = []
    for _ in range(51):  # CosineAnnealingLR(T_max=25, eta_min=lr/lrRatio) table from the reference configuration
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
    np.testing.assert_allclose(lrs, z["cosine_lr"], rtol=1e-9)
    mm = MMETrainingModule(num_cls=4)
    (og, of), (sg, sf) = mm.configure_optimizers()
    assert len(og.param_groups) == 2 and og.param_groups[0]["lr"] == pytest.approx(1e-3 / 3)
    assert og.param_groups[1]["lr"] == pytest.approx(1e-3) and og.param_groups[0]["nesterov"]
    assert sum(p.numel() for p in og.param_groups[1]["params"]) == 4 * 288 + 4


def test_no_cpu_fallback_and_errors():
    m = SimpleTrainModule(num_cls=4)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="GPU only"):
        m.training_step((torch.zeros(1, 3, 64, 64), torch.zeros(1, 64, 64, dtype=torch.long)), 0)
    with pytest.raises(RuntimeError):
        m.featureExtractor.denseBlocksDown[0](torch.zeros(1, 48, 8, 8))  # containers do not run torch ops
    with pytest.raises(AssertionError):
        getClassWeight(torch.tensor([0, 1, 5]), 4)
    w = getClassWeight(torch.tensor([0, 0, 1, 2]), 4)
    assert torch.equal(w, torch.tensor([0.5, 1.0, 1.0, float("inf")]))
    assert float(adentropy(torch.full((1, 4, 2, 2), 0.25))) == pytest.approx(float(np.log(0.25 + 1e-5)), rel=1e-6)


def test_factories_and_aliases():
    assert tiramisu.FCDenseNet57(2).featureExtractor.getFeatureChannels() == 48 + 12 * 4 + 12 * 4 + 12 * 4
    assert tiramisu.FCDenseNet103(12).classifier.finalConv.out_channels == 12
    assert tiramisu.FCDenseNet67Classifier(4).finalConv.in_channels == 288
    net = tiramisu.FCDenseNet67(4)
    assert sum(p.numel() for p in net.parameters()) == 3461220
    x = torch.ones(2, 3, requires_grad=True)
    tiramisu.grad_reverse(x).sum().backward()
    assert torch.equal(x.grad, -torch.ones(2, 3))
    install_aliases()
    from trainingModules.SimpleTrain import SimpleTrainModule as Aliased  # reference import path (train.py:11)
    from models.FCDenseNet.tiramisu import FCDenseNet67Base  # TrainingBase.py:8
    assert Aliased is SimpleTrainModule and FCDenseNet67Base is tiramisu.FCDenseNet67Base


def test_metrics_match_oracle_definitions():
    g = torch.Generator().manual_seed(0)
    probs = torch.softmax(torch.randn(3, 4, 9, 11, generator=g), 1)
    target = torch.randint(0, 3, (3, 9, 11), generator=g)  # class 3 absent from the target
    pred = probs.argmax(1)
    cm = metrics.confusion_matrix(pred, target, 4)
    assert float(metrics.accuracy_from_confusion(cm)) == pytest.approx(float(O.accuracy(pred, target)))
    assert float(metrics.iou_from_confusion(cm)) == pytest.approx(float(O.iou(pred, target)), rel=1e-6)
    assert float(metrics.dice_from_confusion(cm)) == pytest.approx(float(O.dice_score(probs, target)), rel=1e-6)
    pred2 = torch.clamp(pred, max=2)  # class 3 absent from both -> inferred num_classes shrinks
    cm2 = metrics.confusion_matrix(pred2, target, 4)
    assert float(metrics.iou_from_confusion(cm2)) == pytest.approx(float(O.iou(pred2, target)), rel=1e-6)


def test_two_domain_synthetic_batches():
    """BASELINE.json configs[4] workload: the 50/50 source/target mix of TwoDomainDM.train_dataloader
    (dataManagement/dataModules.py:79-85) on synthetic frames of two looks."""
    from sim2real_lane_segment_amd import synthetic as S
    idx, dom = S.two_domain_indices(1000, 100, 20000, 3)
    assert abs(float(dom.float().mean()) - 0.5) < 0.02          # each DOMAIN with probability 1/2, whatever the set sizes
    assert int(idx[dom == 0].max()) < 1000 <= int(idx[dom == 1].min())
    x, y, d = S.make_two_domain_batch(8, 64, 96, seed=5)
    x2, y2, d2 = S.make_two_domain_batch(8, 64, 96, seed=5)
    assert torch.equal(x, x2) and torch.equal(y, y2) and torch.equal(d, d2)
    assert x.shape == (8, 3, 64, 96) and y.shape == (8, 64, 96) and y.dtype == torch.int64
    assert 0 < int(d.sum()) < 8 and int(y.max()) <= 3
    # the two looks have different first-order statistics (darker, greyer target frames)
    assert float(x[d == 1].mean()) < float(x[d == 0].mean()) - 0.1
    # a sample of domain 0 is the plain simulator frame of the same index
    i0 = int(torch.nonzero(d == 0)[0])
    xs, ys = S.make_batch(1, 64, 96, seed=5, first_index=int(S.two_domain_indices(1000, 100, 8, 5)[0][i0]))
    assert torch.equal(xs[0], x[i0]) and torch.equal(ys[0], y[i0])
