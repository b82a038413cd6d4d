#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference's own model and training-step code
(/root/reference/rightLaneNetwork/models/FCDenseNet/{layers,tiramisu}.py and
trainingModules/{TrainingBase,SimpleTrain}.py through a stand-in ``pytorch_lightning`` module --
the real one is not installed), feeds it seeded inputs / deterministic weights / explicit
Dropout2d masks, and stores inputs-by-seed + expected outputs as small ``.npz`` fixtures next
to this script.  Nothing from the reference is copied: fixtures hold numbers only.

    python tests/golden/gen_golden.py            # regenerates every fixture

The weights come from ``oracle.fcdensenet_oracle.init_state`` and are pushed into the reference
classes with ``load_state_dict(strict=True)`` (which also pins state_dict key/shape parity).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/rightLaneNetwork"
sys.path.insert(0, REPO)

from oracle import fcdensenet_oracle as O  # noqa: E402
from tests.golden.common import synth_batch, sample_idx, pack_masks, cfg_to_arrays  # noqa: E402


def _install_lightning_stub():
    """~15-line stand-in so trainingModules/* import verbatim."""
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, **k):
            self.__dict__.setdefault("_logged", {})[name] = value

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    metrics = types.ModuleType("pytorch_lightning.metrics")
    fn = types.ModuleType("pytorch_lightning.metrics.functional")
    fn.accuracy = lambda pred, target: (pred == target).float().mean()
    fn.dice_score = lambda *a, **k: torch.tensor(0.0)
    fn.iou = lambda *a, **k: torch.tensor(0.0)
    metrics.functional = fn
    pl.metrics = metrics
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.metrics"] = metrics
    sys.modules["pytorch_lightning.metrics.functional"] = fn


class DropInjector:
    """Replaces nn.Dropout2d.forward so that masks are explicit inputs."""

    def __init__(self):
        self.scales = None
        self.i = 0
        self._orig = torch.nn.Dropout2d.forward

    def __enter__(self):
        inj = self

        def fwd(mod, x):
            if not mod.training:
                return x
            s = inj.scales[inj.i]
            inj.i += 1
            assert s.shape == x.shape[:2]
            return x * s[:, :, None, None]

        torch.nn.Dropout2d.forward = fwd
        return self

    def __exit__(self, *a):
        torch.nn.Dropout2d.forward = self._orig


def build_reference(cfg: O.NetConfig, st):
    from models.FCDenseNet.tiramisu import FCDenseNetFeatureExtractor, FCDenseNetClassifier
    fe = FCDenseNetFeatureExtractor(in_channels=cfg.in_channels, down_blocks=cfg.down_blocks,
                                    up_blocks=cfg.up_blocks, bottleneck_layers=cfg.bottleneck_layers,
                                    growth_rate=cfg.growth_rate, out_chans_first_conv=cfg.out_chans_first_conv)
    cl = FCDenseNetClassifier(fe.getFeatureChannels(), cfg.n_classes)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.featureExtractor = fe
            self.classifier = cl

        def forward(self, x):
            return self.classifier(self.featureExtractor(x))

    net = Net()
    net.load_state_dict(st, strict=True)
    return net


def gen_small(name, cfg, n, h, w, seed, full=True, absent_class=False):
    """Reduced nets via the reference's own ctor knobs: eval forward, train forward (with masks),
    one training step (loss, acc, grads, running stats, AdamW-updated parameters)."""
    st = O.init_state(cfg, seed)
    x, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    if absent_class:
        y[y == cfg.n_classes - 1] = 0
    scales = O.make_drop_scales(cfg, n, seed + 2)
    out = dict(cfg_to_arrays(cfg), n=n, h=h, w=w, seed=seed, absent_class=int(absent_class))

    net = build_reference(cfg, st)
    net.eval()
    with torch.no_grad():
        feat = net.featureExtractor(x)
        probs_eval = net.classifier(feat)
        logits_eval = net.classifier(feat, useSoftmax=False)
    out["eval_probs"] = probs_eval.numpy()
    out["eval_logits"] = logits_eval.numpy()
    out["eval_feat_sum"] = feat.sum((0, 2, 3)).numpy()
    out["eval_feat_abs"] = feat.abs().sum((0, 2, 3)).numpy()

    # training step through the reference's SimpleTrainModule.training_step, verbatim
    from trainingModules.SimpleTrain import SimpleTrainModule
    mod = SimpleTrainModule.__new__(SimpleTrainModule)
    torch.nn.Module.__init__(mod)
    mod.featureExtractor = net.featureExtractor
    mod.classifier = net.classifier
    mod.lr, mod.decay, mod.lrRatio, mod.num_cls = 1e-3, 1e-4, 1e3, cfg.n_classes
    mod.train()
    captured = []
    hook = mod.classifier.register_forward_hook(lambda m, i, o: captured.append(o.detach().clone()))
    with DropInjector() as inj:
        inj.scales = scales
        loss = mod.training_step((x, y), 0)
        assert inj.i == len(scales)
    hook.remove()
    out["train_loss"] = np.float32(loss.item())
    out["train_acc"] = np.float32(float(mod._logged["tr_acc"]))
    probs_train = captured[0]
    out["train_probs"] = probs_train.numpy() if full else probs_train.numpy()[:, :, ::7, ::5]
    opt, sched = mod.configure_optimizers()
    mod.zero_grad()
    loss.backward()
    named = dict(mod.named_parameters())
    gnorm, gsamp, psamp = {}, {}, {}
    for k, p in named.items():
        g = p.grad.detach()
        if full:
            out["grad/" + k] = g.numpy().copy()
        else:
            idx = sample_idx(g.numel(), 64, 1234)
            out["gradnorm/" + k] = np.float32(g.norm().item())
            out["gradsum/" + k] = np.float32(g.double().sum().item())
            out["gradsamp/" + k] = g.reshape(-1)[idx].numpy().copy()
    opt[0].step()
    for k, p in named.items():
        if full:
            out["param1/" + k] = p.detach().numpy().copy()
        else:
            idx = sample_idx(p.numel(), 64, 1234)
            out["param1samp/" + k] = p.detach().reshape(-1)[idx].numpy().copy()
    for k, b in mod.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf1/" + k] = b.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "acc", float(mod._logged["tr_acc"]))


def gen_fcd67_eval(name, n, h, w, seed):
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, seed)
    x, _ = synth_batch(n, h, w, 4, seed + 1)
    net = build_reference(cfg, st)
    net.eval()
    with torch.no_grad():
        feat = net.featureExtractor(x)
        logits = net.classifier(feat, useSoftmax=False)
        probs = net.classifier(feat)
    mask = torch.max(probs, 1)[1]
    top2 = logits.topk(2, dim=1)[0]
    gap = (top2[:, 0] - top2[:, 1])
    idx = sample_idx(n * h * w, 1024, 99)
    out = dict(cfg_to_arrays(cfg), n=n, h=h, w=w, seed=seed,
               mask_packed=pack_masks(mask, 4),
               gap_min=np.float32(gap.min().item()),
               near_tie_idx=torch.nonzero(gap.reshape(-1) < 2e-3).reshape(-1).numpy().astype(np.int64),
               sample_idx=idx.numpy().astype(np.int64),
               probs_samp=probs.permute(0, 2, 3, 1).reshape(-1, 4)[idx].numpy(),
               logits_samp=logits.permute(0, 2, 3, 1).reshape(-1, 4)[idx].numpy(),
               feat_sum=feat.double().sum((0, 2, 3)).float().numpy(),
               feat_abs=feat.double().abs().sum((0, 2, 3)).float().numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "gap_min", float(gap.min()), "near ties", len(out["near_tie_idx"]))


def gen_fcd67_train(name, n, h, w, seed, steps):
    """FCDenseNet67 training steps through the reference's training_step + its configured AdamW."""
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, seed)
    net = build_reference(cfg, st)
    from trainingModules.SimpleTrain import SimpleTrainModule
    mod = SimpleTrainModule.__new__(SimpleTrainModule)
    torch.nn.Module.__init__(mod)
    mod.featureExtractor, mod.classifier = net.featureExtractor, net.classifier
    mod.lr, mod.decay, mod.lrRatio, mod.num_cls = 1e-3, 1e-4, 1e3, 4
    mod.train()
    opt, _ = mod.configure_optimizers()
    out = dict(cfg_to_arrays(cfg), n=n, h=h, w=w, seed=seed, steps=steps)
    losses, accs = [], []
    named = dict(mod.named_parameters())
    for s in range(steps):
        x, y = synth_batch(n, h, w, 4, seed + 10 * s + 1)
        if s == 0:
            y[0][y[0] == 3] = 0  # sample 0 has no class 3; batch still has it
        scales = O.make_drop_scales(cfg, n, seed + 10 * s + 2)
        with DropInjector() as inj:
            inj.scales = scales
            loss = mod.training_step((x, y), s)
        mod.zero_grad()
        loss.backward()
        if s == 0:
            # per tensor: L2 norm, sum, and up to 1024 sampled entries (tensors with <= 1024 elements are stored whole):
            # the tests estimate the L2-relative error of every tensor and of the whole arena from these
            for k, p in named.items():
                g = p.grad.detach()
                idx = sample_idx(g.numel(), 1024, 1234)
                out["gradnorm/" + k] = np.float32(g.norm().item())
                out["gradsum/" + k] = np.float32(g.double().sum().item())
                out["gradsamp/" + k] = g.reshape(-1)[idx].numpy().copy()
        opt[0].step()
        if s == 0:
            for k, p in named.items():
                idx = sample_idx(p.numel(), 64, 1234)
                out["param1samp/" + k] = p.detach().reshape(-1)[idx].numpy().copy()
            for k, b in mod.named_buffers():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    out["buf1/" + k] = b.numpy().copy()
        losses.append(loss.item())
        accs.append(float(mod._logged["tr_acc"]))
        print(name, "step", s, "loss", losses[-1], "acc", accs[-1], flush=True)
    out["losses"] = np.array(losses, np.float32)
    out["accs"] = np.array(accs, np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def gen_mme(name, cfg, n, h, w, seed):
    """MMETrainingModule.training_step, verbatim, for both optimizer indices on a reduced net, followed by one
    step of the optimizers the reference configures (SGD-nesterov with two lr groups, AdamW)."""
    from trainingModules.MMETrainingModule import MMETrainingModule
    st = O.init_state(cfg, seed)
    net = build_reference(cfg, st)
    mod = MMETrainingModule.__new__(MMETrainingModule)
    torch.nn.Module.__init__(mod)
    mod.featureExtractor, mod.classifier = net.featureExtractor, net.classifier
    mod.lr, mod.decay, mod.lrRatio, mod.num_cls = 1e-3, 1e-4, 1e3, cfg.n_classes
    mod.train()
    (optG, optF), _ = mod.configure_optimizers()
    xl, y = synth_batch(n, h, w, cfg.n_classes, seed + 1)
    xu, _ = synth_batch(n, h, w, cfg.n_classes, seed + 5)
    out = dict(cfg_to_arrays(cfg), n=n, h=h, w=w, seed=seed)
    named = dict(mod.named_parameters())
    # optimizer_idx 0: unlabelled -> grad_reverse -> classifier -> adentropy(0.1)
    scales0 = O.make_drop_scales(cfg, n, seed + 2)
    with DropInjector() as inj:
        inj.scales = scales0
        loss0 = mod.training_step((xl, xu, y, None), 0, optimizer_idx=0)
    mod.zero_grad()
    loss0.backward()
    out["loss0"] = np.float32(loss0.item())
    for k, p in named.items():
        out["grad0/" + k] = p.grad.detach().numpy().copy()
    optG.step()
    for k, p in named.items():
        out["param_after_sgd/" + k] = p.detach().numpy().copy()
    # second SGD step with the same gradients exercises the momentum buffer
    optG.step()
    for k, p in named.items():
        out["param_after_sgd2/" + k] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss0", float(loss0))


def gen_encdec(name, n_feat, n_levels, k, n_lin, n, h, w, seed):
    """The reference's legacy EncDecNet (models/EncDecNet.py): eval forward and train-mode forward with injected
    Dropout masks; default-initialised weights are stored in the fixture (small nets)."""
    from models.EncDecNet import EncDecNet
    from oracle import encdecnet_oracle as E
    torch.manual_seed(seed)
    net = EncDecNet(n_feat, n_levels, k, n_lin)
    with torch.no_grad():  # perturb BN so that the affine part and running stats are exercised
        g = torch.Generator().manual_seed(seed + 1)
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.75 + 0.5 * torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * (torch.rand(m.bias.shape, generator=g) - 0.5))
                m.running_mean.copy_(0.1 * (torch.rand(m.bias.shape, generator=g) - 0.5))
                m.running_var.copy_(0.75 + 0.5 * torch.rand(m.bias.shape, generator=g))
    out = dict(n_feat=n_feat, n_levels=n_levels, k=k, n_lin=n_lin, n=n, h=h, w=w, seed=seed)
    for key, v in net.state_dict().items():
        out["state/" + key] = v.numpy().copy()
    x, _ = synth_batch(n, h, w, 2, seed + 2)
    net.eval()
    with torch.no_grad():
        out["eval_out"] = net(x).numpy()
    masks = E.make_masks(E.mask_shapes(n, h, w, n_feat, n_levels, k), 0.3, seed + 3)
    state = {"i": 0}
    orig = torch.nn.Dropout.forward

    def fwd(mod, t):
        if not mod.training:
            return t
        m = masks[state["i"]]
        state["i"] += 1
        assert m.shape == t.shape
        return t * m

    torch.nn.Dropout.forward = fwd
    try:
        net.train()
        with torch.no_grad():
            out["train_out"] = net(x).numpy()
        assert state["i"] == len(masks)
    finally:
        torch.nn.Dropout.forward = orig
    for key, v in net.state_dict().items():
        if "running" in key:
            out["buf1/" + key] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", out["eval_out"].shape)


def gen_encdec_refsize(name, k, seed):
    """EncDecNet at the sizes the reference itself builds (models/EncDecNet.py:119-130: EncDecNet(64, 3, 7) on
    ones(1, 3, 120, 160); (64, 3, 3) is the survey's second size): eval forward of the reference class with the
    deterministic weights of tests/golden/common.py:det_state on the script's all-ones input and on a seeded frame."""
    from models.EncDecNet import EncDecNet
    from tests.golden.common import det_state
    net = EncDecNet(64, 3, k)
    assert net.getNParams() == (7237570 if k == 7 else 1331650)
    net.load_state_dict(det_state({key: v.shape for key, v in net.state_dict().items()}, seed), strict=True)
    net.eval()
    x, _ = synth_batch(1, 120, 160, 2, seed + 2)
    out = dict(k=k, seed=seed)
    with torch.no_grad():
        for tag, inp in (("ones", torch.ones(1, 3, 120, 160)), ("rand", x)):
            y = net(inp)
            assert y.shape == (1, 2, 120, 160)
            idx = sample_idx(y.numel(), 4096, 77)
            out[tag + "_samp"] = y.reshape(-1)[idx].numpy()
            out[tag + "_mask"] = pack_masks(y.argmax(1), 2)
            out[tag + "_gapmin"] = float((y[:, 0] - y[:, 1]).abs().min())
            out[tag + "_near"] = torch.nonzero(((y[:, 0] - y[:, 1]).abs() < 1e-4).reshape(-1)).reshape(-1).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", out["rand_gapmin"], len(out["rand_near"]))


def gen_misc(name):
    """Third-party arithmetic at the reference's call sites: AdamW(lr,weight_decay) 3 steps,
    CosineAnnealingLR(25, eta_min=lr/lrRatio) table, getClassWeight incl. an absent class."""
    from trainingModules.TrainingBase import getClassWeight
    gen = torch.Generator().manual_seed(7)
    p = torch.nn.Parameter(torch.randn(1000, generator=gen))
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 25, eta_min=1e-3 / 1e3)
    out = {"adamw_p0": p.detach().numpy().copy()}
    gs = []
    for s in range(3):
        g = torch.randn(1000, generator=gen)
        gs.append(g.numpy().copy())
        p.grad = g.clone()
        opt.step()
        out[f"adamw_p{s + 1}"] = p.detach().numpy().copy()
    out["adamw_grads"] = np.stack(gs)
    lrs = []
    for e in range(51):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    out["cosine_lr"] = np.array(lrs, np.float64)
    y = torch.randint(0, 3, (2, 12, 16), generator=gen)  # class 3 absent
    out["cw_targets"] = y.numpy()
    out["cw_weights"] = getClassWeight(y, 4).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok")


def main():
    assert os.path.isdir(REF), "reference tree not present: fixtures can only be generated in the build container"
    sys.path.insert(0, REF)
    _install_lightning_stub()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["misc", "mme", "encdec", "tiny", "tiny_odd", "g16", "g16_absent", "fcd67_eval", "fcd67_eval480",
                             "fcd67_train"]
    tiny = O.NetConfig(down_blocks=(2, 2), up_blocks=(2, 2), bottleneck_layers=2, growth_rate=4,
                       out_chans_first_conv=8, n_classes=4)
    g16 = O.NetConfig(down_blocks=(2, 2), up_blocks=(2, 2), bottleneck_layers=2, growth_rate=16,
                      out_chans_first_conv=48, n_classes=4)
    if "misc" in which:
        gen_misc("misc")
    if "encdec" in which:
        gen_encdec("encdec_k3_relu_24x40", 8, 2, 3, "relu", 2, 24, 40, 900)
        gen_encdec("encdec_k7_leaky_40x56", 8, 3, 7, "leakyRelu", 2, 40, 56, 910)
        gen_encdec("encdec_k3_prelu_30x34", 6, 2, 3, "prelu", 1, 30, 34, 920)
    if "encdec_ref" in which:
        gen_encdec_refsize("encdec_ref_64_3_3_120x160", 3, 930)
        gen_encdec_refsize("encdec_ref_64_3_7_120x160", 7, 940)
    if "mme" in which:
        gen_mme("mme_tiny_40x56", tiny, 2, 40, 56, 800)
    if "tiny" in which:
        gen_small("tiny_40x56", tiny, 2, 40, 56, 100, full=True)
    if "tiny_odd" in which:
        gen_small("tiny_33x47", tiny, 3, 33, 47, 200, full=True)
    if "g16" in which:
        gen_small("g16_32x48", g16, 2, 32, 48, 300, full=False)
    if "g16_absent" in which:
        gen_small("g16_absent_30x34", g16, 1, 30, 34, 400, full=False, absent_class=True)
    if "fcd67_eval" in which:
        gen_fcd67_eval("fcd67_eval_120x160", 2, 120, 160, 500)
    if "fcd67_eval480" in which:
        gen_fcd67_eval("fcd67_eval_480x640", 1, 480, 640, 600)
    if "fcd67_train" in which:
        gen_fcd67_train("fcd67_train_120x160", 2, 120, 160, 700, steps=5)


if __name__ == "__main__":
    main()
