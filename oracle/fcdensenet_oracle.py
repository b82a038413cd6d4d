"""CPU oracle for the lane-segmentation hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain-PyTorch, CPU, fp32 restatement of the reference algorithm.
It is *not* part of the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``sim2real_lane_segment_amd``) never routes through it and fails loudly when
the HIP library is missing.

Parity pin: ``tests/golden/*.npz`` were produced by ``tests/golden/gen_golden.py``
which imports the reference's own model files (models/FCDenseNet/{layers,tiramisu}.py
and trainingModules/SimpleTrain.py via a stub ``pytorch_lightning``) in the build
container; ``tests/test_oracle_golden.py`` checks this restatement against them.

Every function cites the reference lines it restates (paths relative to
``/root/reference/rightLaneNetwork``).

State is a flat ``dict[str, Tensor]`` using the reference's ``state_dict`` keys
(``featureExtractor.*`` / ``classifier.*``).  Dropout2d masks are explicit
inputs (``drop_scales``: list of ``[N, C]`` tensors holding 0 or 1/(1-p)), in
module execution order, so both sides of a parity check see identical masks.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class NetConfig:
    """Constructor knobs of FCDenseNetFeatureExtractor (models/FCDenseNet/tiramisu.py:21-24)."""
    in_channels: int = 3
    down_blocks: Tuple[int, ...] = (5, 5, 5, 5, 5)
    up_blocks: Tuple[int, ...] = (5, 5, 5, 5, 5)
    bottleneck_layers: int = 5
    growth_rate: int = 16
    out_chans_first_conv: int = 48
    n_classes: int = 4
    temperature: float = 0.05
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    drop_p: float = 0.2


def fcdensenet67_config(n_classes: int = 4) -> NetConfig:
    """models/FCDenseNet/tiramisu.py:185-194 (FCDenseNet67Base / FCDenseNet67Classifier)."""
    return NetConfig(n_classes=n_classes)


# --------------------------------------------------------------------------
# parameter inventory (shapes follow the reference constructors)
# --------------------------------------------------------------------------

def _bn_entries(prefix: str, c: int):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def state_spec(cfg: NetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Keys and shapes of ``TrainingBase(...).state_dict()`` in registration order.

    Follows tiramisu.py:21-87 (feature extractor ctor), layers.py:5-12,18-24,43-52,58-63,72-76
    and tiramisu.py:112-118 (classifier).
    """
    g = cfg.growth_rate
    out: List[Tuple[str, Tuple[int, ...]]] = []
    fe = "featureExtractor."
    out += [(fe + "firstconv.weight", (cfg.out_chans_first_conv, cfg.in_channels, 3, 3)),
            (fe + "firstconv.bias", (cfg.out_chans_first_conv,))]
    cur = cfg.out_chans_first_conv
    skips = []
    dense_down, td = [], []
    for i, nl in enumerate(cfg.down_blocks):
        for j in range(nl):
            cin = cur + j * g
            p = f"{fe}denseBlocksDown.{i}.layers.{j}."
            dense_down += _bn_entries(p + "norm", cin)
            dense_down += [(p + "conv.weight", (g, cin, 3, 3)), (p + "conv.bias", (g,))]
        cur += g * nl
        skips.insert(0, cur)
        p = f"{fe}transDownBlocks.{i}."
        td += _bn_entries(p + "norm", cur)
        td += [(p + "conv.weight", (cur, cur, 1, 1)), (p + "conv.bias", (cur,))]
    out += dense_down + td
    for j in range(cfg.bottleneck_layers):
        cin = cur + j * g
        p = f"{fe}bottleneck.bottleneck.layers.{j}."
        out += _bn_entries(p + "norm", cin)
        out += [(p + "conv.weight", (g, cin, 3, 3)), (p + "conv.bias", (g,))]
    prev = g * cfg.bottleneck_layers
    tu, dense_up = [], []
    for i, nl in enumerate(cfg.up_blocks):
        p = f"{fe}transUpBlocks.{i}.convTrans."
        tu += [(p + "weight", (prev, prev, 3, 3)), (p + "bias", (prev,))]
        cur = prev + skips[i]
        for j in range(nl):
            cin = cur + j * g
            p = f"{fe}denseBlocksUp.{i}.layers.{j}."
            dense_up += _bn_entries(p + "norm", cin)
            dense_up += [(p + "conv.weight", (g, cin, 3, 3)), (p + "conv.bias", (g,))]
        prev = g * nl
        cur += prev
    out += tu + dense_up
    out += [("classifier.finalConv.weight", (cfg.n_classes, cur, 1, 1)),
            ("classifier.finalConv.bias", (cfg.n_classes,))]
    return out


def feature_channels(cfg: NetConfig) -> int:
    """tiramisu.py:87,108-109 (``featureChannels``)."""
    return state_spec(cfg)[-2][1][1]


def dropout_channels(cfg: NetConfig) -> List[int]:
    """Channel count of every Dropout2d call in forward execution order
    (layers.py:12,51 as visited by tiramisu.py:89-102)."""
    g = cfg.growth_rate
    chans: List[int] = []
    cur = cfg.out_chans_first_conv
    for nl in cfg.down_blocks:
        chans += [g] * nl
        cur += g * nl
        chans.append(cur)
    chans += [g] * cfg.bottleneck_layers
    for nl in cfg.up_blocks:
        chans += [g] * nl
    return chans


def init_state(cfg: NetConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic initialiser used by fixtures and tests (the reference has no custom
    init; this mimics PyTorch's default scale: uniform(+-1/sqrt(fan_in)) for conv weight and
    bias, BN gamma near 1 / beta near 0 but perturbed so that affine terms are exercised)."""
    gen = torch.Generator().manual_seed(seed)
    st: Dict[str, Tensor] = {}
    for name, shape in state_spec(cfg):
        if name.endswith("num_batches_tracked"):
            st[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            st[name] = 0.1 * (torch.rand(shape, generator=gen) - 0.5)
        elif name.endswith("running_var"):
            st[name] = 0.75 + 0.5 * torch.rand(shape, generator=gen)
        elif ".norm.weight" in name:
            st[name] = 0.75 + 0.5 * torch.rand(shape, generator=gen)
        elif ".norm.bias" in name:
            st[name] = 0.2 * (torch.rand(shape, generator=gen) - 0.5)
        elif name.endswith("weight"):
            if "convTrans" in name:
                fan_in = shape[1] * shape[2] * shape[3]
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            st[name] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
        else:  # conv bias
            st[name] = (torch.rand(shape, generator=gen) * 2 - 1) * 0.05
    return st


def make_drop_scales(cfg: NetConfig, n: int, seed: int) -> List[Tensor]:
    """Dropout2d(0.2) masks as explicit inputs: per (sample, channel) 0 or 1/(1-p)
    (layers.py:12,51; torch Dropout2d zeroes whole channels and rescales survivors)."""
    gen = torch.Generator().manual_seed(seed)
    keep = 1.0 - cfg.drop_p
    return [(torch.rand(n, c, generator=gen) < keep).float() / keep for c in dropout_channels(cfg)]


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------

class _DropFeed:
    def __init__(self, scales: Optional[Sequence[Tensor]]):
        self.scales = scales
        self.i = 0

    def __call__(self, x: Tensor) -> Tensor:
        if self.scales is None:
            return x
        s = self.scales[self.i]
        self.i += 1
        assert s.shape == x.shape[:2], (s.shape, x.shape)
        return x * s[:, :, None, None]


def _bn(st, prefix, x, training, cfg, new_stats):
    """nn.BatchNorm2d as used by layers.py:8,46 (train: batch statistics, biased variance for
    normalisation, unbiased for the running estimate, momentum 0.1)."""
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, cfg.bn_momentum, cfg.bn_eps)
        if new_stats is not None:
            new_stats[prefix + ".running_mean"] = rm2
            new_stats[prefix + ".running_var"] = rv2
            new_stats[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, rm, rv, w, b, False, cfg.bn_momentum, cfg.bn_eps)


def _dense_layer(st, prefix, x, training, cfg, drop, new_stats):
    """layers.py:5-15: BN -> ReLU -> Conv3x3(pad 1, bias) -> Dropout2d."""
    y = F.relu(_bn(st, prefix + ".norm", x, training, cfg, new_stats))
    y = F.conv2d(y, st[prefix + ".conv.weight"], st[prefix + ".conv.bias"], stride=1, padding=1)
    return drop(y) if training else y


def _dense_block(st, prefix, x, n_layers, upsample, training, cfg, drop, new_stats):
    """layers.py:18-40."""
    new = []
    for j in range(n_layers):
        out = _dense_layer(st, f"{prefix}.layers.{j}", x, training, cfg, drop, new_stats)
        x = torch.cat([x, out], 1)
        new.append(out)
    return torch.cat(new, 1) if upsample else x


def _transition_down(st, prefix, x, training, cfg, drop, new_stats):
    """layers.py:43-55: BN -> ReLU -> Conv1x1 -> Dropout2d -> MaxPool2d(2)."""
    y = F.relu(_bn(st, prefix + ".norm", x, training, cfg, new_stats))
    y = F.conv2d(y, st[prefix + ".conv.weight"], st[prefix + ".conv.bias"])
    if training:
        y = drop(y)
    return F.max_pool2d(y, 2)


def _transition_up(st, prefix, x, skip):
    """layers.py:58-69,82-86: ConvTranspose2d(k3,s2,p0) -> center_crop -> cat([out, skip])."""
    out = F.conv_transpose2d(x, st[prefix + ".convTrans.weight"], st[prefix + ".convTrans.bias"], stride=2)
    _, _, h, w = out.shape
    mh, mw = skip.shape[2], skip.shape[3]
    xy1 = (w - mw) // 2
    xy2 = (h - mh) // 2
    out = out[:, :, xy2:xy2 + mh, xy1:xy1 + mw]
    return torch.cat([out, skip], 1)


def features_forward(st: Dict[str, Tensor], x: Tensor, cfg: NetConfig, training: bool = False,
                     drop_scales: Optional[Sequence[Tensor]] = None,
                     new_stats: Optional[Dict[str, Tensor]] = None,
                     normalize: bool = True) -> Tensor:
    """FCDenseNetFeatureExtractor.forward (tiramisu.py:89-106)."""
    fe = "featureExtractor"
    drop = _DropFeed(drop_scales if training else None)
    out = F.conv2d(x, st[fe + ".firstconv.weight"], st[fe + ".firstconv.bias"], stride=1, padding=1)
    skips = []
    for i, nl in enumerate(cfg.down_blocks):
        out = _dense_block(st, f"{fe}.denseBlocksDown.{i}", out, nl, False, training, cfg, drop, new_stats)
        skips.append(out)
        out = _transition_down(st, f"{fe}.transDownBlocks.{i}", out, training, cfg, drop, new_stats)
    out = _dense_block(st, f"{fe}.bottleneck.bottleneck", out, cfg.bottleneck_layers, True, training, cfg, drop,
                       new_stats)
    for i, nl in enumerate(cfg.up_blocks):
        skip = skips.pop()
        out = _transition_up(st, f"{fe}.transUpBlocks.{i}", out, skip)
        out = _dense_block(st, f"{fe}.denseBlocksUp.{i}", out, nl, i < len(cfg.up_blocks) - 1, training, cfg, drop,
                           new_stats)
    if training and drop_scales is not None:
        assert drop.i == len(drop_scales)
    return F.normalize(out) if normalize else out


def classifier_forward(st: Dict[str, Tensor], feat: Tensor, cfg: NetConfig, use_softmax: bool = True) -> Tensor:
    """FCDenseNetClassifier.forward (tiramisu.py:120-125): 1x1 conv, / T, softmax(dim=1)."""
    z = F.conv2d(feat, st["classifier.finalConv.weight"], st["classifier.finalConv.bias"])
    z = z / cfg.temperature
    return F.softmax(z, dim=1) if use_softmax else z


def forward(st, x, cfg, training=False, drop_scales=None, new_stats=None, use_softmax=True) -> Tensor:
    """TrainingBase.forward (trainingModules/TrainingBase.py:54-57)."""
    return classifier_forward(st, features_forward(st, x, cfg, training, drop_scales, new_stats), cfg, use_softmax)


# --------------------------------------------------------------------------
# loss / metrics
# --------------------------------------------------------------------------

def get_class_weight(targets: Tensor, max_classes: Optional[int] = None) -> Tensor:
    """getClassWeight (trainingModules/TrainingBase.py:12-23): reciprocal of per-class pixel
    counts; absent classes get inf (never indexed by the loss)."""
    elements, counts = torch.unique(targets, sorted=True, return_counts=True)
    if max_classes:
        assert max_classes > int(elements.max()), \
            f"Found more label classes than given maxClasses={max_classes}"
    else:
        max_classes = int(elements.max()) + 1
    cnt = torch.zeros(max_classes, dtype=torch.float)
    for idx, c in zip(elements, counts):
        cnt[idx] = c
    return torch.reciprocal(cnt)


def training_loss(probs: Tensor, y: Tensor, n_classes: int) -> Tuple[Tensor, Tensor]:
    """SimpleTrainModule.training_step loss + accuracy (trainingModules/SimpleTrain.py:15-20):
    class-weighted cross_entropy applied to the softmax *probabilities* (double softmax),
    accuracy(argmax, y) * 100."""
    loss = F.cross_entropy(probs, y, weight=get_class_weight(y, n_classes).to(probs.dtype))
    labels_hat = torch.max(probs, 1)[1]
    acc = (labels_hat == y).float().mean() * 100
    return loss, acc


def accuracy(pred: Tensor, target: Tensor) -> Tensor:
    """pytorch_lightning 1.2.1 metrics.functional.accuracy on label tensors = match rate
    (third-party, absent here: restated from its published definition; parity unpinned)."""
    return (pred == target).float().mean()


def confusion(pred: Tensor, target: Tensor, num_classes: int) -> Tensor:
    idx = target.reshape(-1) * num_classes + pred.reshape(-1)
    return torch.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def iou(pred: Tensor, target: Tensor, num_classes: Optional[int] = None, absent_score: float = 0.0) -> Tensor:
    """pytorch_lightning 1.2.1 metrics.functional.iou (third-party, absent here; restated from
    its published definition, parity unpinned): mean over classes of diag/(row+col-diag) of the
    confusion matrix, num_classes inferred as max(pred,target)+1, classes absent from both
    pred and target score ``absent_score``."""
    if num_classes is None:
        num_classes = int(max(pred.max(), target.max())) + 1
    cm = confusion(pred, target, num_classes).double()
    inter = torch.diag(cm)
    union = cm.sum(0) + cm.sum(1) - inter
    scores = torch.where(union > 0, inter / union.clamp(min=1), torch.full_like(inter, absent_score))
    return scores.mean().float()


def dice_score(probs: Tensor, target: Tensor, bg: bool = False, nan_score: float = 0.0,
               no_fg_score: float = 0.0) -> Tensor:
    """pytorch_lightning 1.2.1 metrics.functional.dice_score (third-party, absent here;
    restated from its published definition, parity unpinned): mean over classes (1..C-1 unless
    bg) of 2TP/(2TP+FP+FN) on argmax; classes absent from target score ``no_fg_score``."""
    n_cls = probs.shape[1]
    pred = probs.argmax(1)
    start = 0 if bg else 1
    scores = []
    for c in range(start, n_cls):
        if not (target == c).any():
            scores.append(torch.tensor(no_fg_score))
            continue
        tp = ((pred == c) & (target == c)).sum().float()
        fp = ((pred == c) & (target != c)).sum().float()
        fn = ((pred != c) & (target == c)).sum().float()
        denom = 2 * tp + fp + fn
        scores.append(2 * tp / denom if denom > 0 else torch.tensor(nan_score))
    return torch.stack(scores).mean()


def evaluate_batch(st, x, y, cfg):
    """TrainingBase.evaluate_batch (trainingModules/TrainingBase.py:79-96): eval forward,
    *unweighted* CE on probabilities, argmax, accuracy/dice/iou each times batch size."""
    probs = forward(st, x, cfg, training=False)
    loss = F.cross_entropy(probs, y)
    labels_hat = torch.max(probs, 1)[1]
    w = x.shape[0]
    return {"loss": loss * w, "acc": accuracy(labels_hat, y) * w, "dice": dice_score(probs, y) * w,
            "iou": iou(labels_hat, y) * w, "weight": w}


# --------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------

def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-4) -> None:
    """torch.optim.AdamW single-tensor update as configured by SimpleTrain.py:27-28
    (decoupled decay, no amsgrad).  In place on p, m, v; ``step`` is 1-based."""
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def cosine_lr(epoch: int, base_lr: float, lr_ratio: float, t_max: int = 25) -> float:
    """CosineAnnealingLR(optimizer, 25, eta_min=lr/lrRatio) closed form (SimpleTrain.py:29)."""
    eta_min = base_lr / lr_ratio
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


class _GradReverse(torch.autograd.Function):
    """GradReverse (models/FCDenseNet/tiramisu.py:7-18): identity forward, negated gradient."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.neg()


def adentropy(probs: Tensor, lamda: float = 1.0) -> Tensor:
    """trainingModules/MMETrainingModule.py:10-11."""
    return lamda * torch.mean(torch.sum(probs * torch.log(probs + 1e-5), 1))


def mme_unlabelled_step(st: Dict[str, Tensor], x: Tensor, cfg: NetConfig, drop_scales, lamda: float = 0.1):
    """MMETrainingModule.training_step with optimizer_idx == 0 (MMETrainingModule.py:28-33): features ->
    grad_reverse -> classifier -> adentropy(lamda).  Returns (loss, grads dict)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in st.items() if is_param(k)}
    work = dict(st)
    work.update(params)
    feat = features_forward(work, x, cfg, True, drop_scales, {})
    probs = classifier_forward(work, _GradReverse.apply(feat), cfg)
    loss = adentropy(probs, lamda)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), {k: g for k, g in zip(params.keys(), grads)}


def sgd_nesterov_step(p: Tensor, g: Tensor, buf: Optional[Tensor], lr: float, momentum: float = 0.9,
                      weight_decay: float = 1e-4):
    """torch.optim.SGD(momentum, nesterov=True, dampening=0) single-tensor update as configured by
    MMETrainingModule.py:17-20.  Returns the new momentum buffer."""
    g = g + weight_decay * p
    buf = g.clone() if buf is None else buf.mul(momentum).add(g)
    p.sub_(lr * (g + momentum * buf))
    return buf


def is_param(name: str) -> bool:
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked"))


@dataclass
class TrainState:
    st: Dict[str, Tensor]
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)
    step: int = 0


def train_step(ts: TrainState, x: Tensor, y: Tensor, cfg: NetConfig, drop_scales, lr: float = 1e-3,
               weight_decay: float = 1e-4, apply_update: bool = True):
    """One SimpleTrainModule step: training_step (SimpleTrain.py:11-25) + backward + AdamW
    (SimpleTrain.py:27-28).  Returns (loss, acc, grads dict, probs)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in ts.st.items() if is_param(k)}
    work = dict(ts.st)
    work.update(params)
    new_stats: Dict[str, Tensor] = {}
    probs = forward(work, x, cfg, training=True, drop_scales=drop_scales, new_stats=new_stats)
    loss, acc = training_loss(probs, y, cfg.n_classes)
    grads = torch.autograd.grad(loss, list(params.values()))
    gd = {k: g for k, g in zip(params.keys(), grads)}
    for k, v in new_stats.items():
        ts.st[k] = v.detach()
    if apply_update:
        ts.step += 1
        for k in params:
            if k not in ts.m:
                ts.m[k] = torch.zeros_like(ts.st[k])
                ts.v[k] = torch.zeros_like(ts.st[k])
            adamw_step(ts.st[k], gd[k], ts.m[k], ts.v[k], ts.step, lr, weight_decay=weight_decay)
    return loss.detach(), acc.detach(), gd, probs.detach()
