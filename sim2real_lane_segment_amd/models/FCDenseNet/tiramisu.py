"""FC-DenseNet ("Tiramisu") on the MI355X HIP path.

API mirror of rightLaneNetwork/models/FCDenseNet/tiramisu.py: same classes, constructor
arguments, factory functions and state_dict keys (tiramisu.py:7-18,21-125,128-194).  The modules
hold parameters; ``forward`` hands the whole network to librln.so (see ../../engine.py).

Autograd boundary: the fused training steps and the fused ``forward`` of ``FCDenseNet`` / ``TrainingBase`` (train mode,
autograd enabled) are differentiable end to end inside the HIP library; ``forward`` of the stand-alone pieces
(``FCDenseNetFeatureExtractor``, ``FCDenseNetClassifier``) is inference / evaluation only and raises when it is asked
for gradients.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ...engine import classifier_op
from ...owner import EngineOwner, ForwardFn
from .layers import Bottleneck, DenseBlock, TransitionDown, TransitionUp  # noqa: F401  (re-exported like the reference)
from .layers import DenseLayer, center_crop  # noqa: F401


class GradReverse(Function):
    """Identity forward, negated gradient (tiramisu.py:7-14)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg()


def grad_reverse(x):
    return GradReverse.apply(x)


class FCDenseNetFeatureExtractor(nn.Module, EngineOwner):
    """tiramisu.py:21-109.  Output: per-pixel L2-normalised ``featureChannels`` features."""

    def __init__(self, in_channels=3, down_blocks=(5, 5, 5, 5, 5), up_blocks=(5, 5, 5, 5, 5), bottleneck_layers=5,
                 growth_rate=16, out_chans_first_conv=48):
        super().__init__()
        if len(down_blocks) != len(up_blocks):
            raise ValueError("down_blocks and up_blocks must have the same length")
        self.in_channels = in_channels
        self.down_blocks = tuple(down_blocks)
        self.up_blocks = tuple(up_blocks)
        self.bottleneck_layers = bottleneck_layers
        self.growth_rate = growth_rate
        self.out_chans_first_conv = out_chans_first_conv

        self.firstconv = nn.Conv2d(in_channels, out_chans_first_conv, kernel_size=3, stride=1, padding=1, bias=True)
        channels = out_chans_first_conv
        skip_channels = []
        self.denseBlocksDown = nn.ModuleList()
        self.transDownBlocks = nn.ModuleList()
        for n_layers in self.down_blocks:
            self.denseBlocksDown.append(DenseBlock(channels, growth_rate, n_layers))
            channels += growth_rate * n_layers
            skip_channels.append(channels)
            self.transDownBlocks.append(TransitionDown(channels))
        self.bottleneck = Bottleneck(channels, growth_rate, bottleneck_layers)
        carried = growth_rate * bottleneck_layers
        self.transUpBlocks = nn.ModuleList()
        self.denseBlocksUp = nn.ModuleList()
        last = len(self.up_blocks) - 1
        for i, n_layers in enumerate(self.up_blocks):
            self.transUpBlocks.append(TransitionUp(carried, carried))
            channels = carried + skip_channels[-1 - i]
            self.denseBlocksUp.append(DenseBlock(channels, growth_rate, n_layers, upsample=(i != last)))
            carried = growth_rate * n_layers
            channels += carried
        self.featureChannels = channels
        self._rln_parent = None  # set by an enclosing owner (FCDenseNet / TrainingBase)

    # nn.Module API: moving the module invalidates the arena aliases; they are rebuilt lazily
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._rln_mark_dirty()
        parent = self.__dict__.get("_rln_parent")
        if parent is not None:
            parent._rln_mark_dirty()
        return out

    @property
    def featureExtractor(self):  # EngineOwner protocol when used stand-alone
        return self

    def getFeatureChannels(self):
        return self.featureChannels

    def forward(self, x):
        """Stand-alone feature extractor: inference / evaluation only.  Differentiable use goes through the owning
        module (``TrainingBase.forward`` / ``FCDenseNet.forward`` / the fused training steps): composing
        ``featureExtractor -> grad_reverse -> classifier`` by hand (MMETrainingModule.py:28-33) is what
        ``MMETrainingModule.training_step`` runs fused; asking THIS call for gradients raises instead of silently
        returning a tensor without grad_fn."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError(
                "FCDenseNetFeatureExtractor.forward is inference-only on the HIP path: call it under torch.no_grad() / "
                "in eval mode, or differentiate through the owning module (TrainingBase.forward, FCDenseNet.forward, "
                "SimpleTrainModule / MMETrainingModule.training_step)")
        parent = self.__dict__.get("_rln_parent")
        owner = parent if parent is not None else self
        eng = owner._rln_sync()
        with torch.no_grad():
            _, feat = eng.forward(x, training=self.training, with_backward=False, want_probs=False, want_feat=True)
        return feat


class FCDenseNetClassifier(nn.Module):
    """tiramisu.py:112-125: Conv2d(in, n_classes, k) -> / T -> Softmax(dim=1)."""

    def __init__(self, in_channels, n_classes, temperature=0.05, kernel_size=1):
        super().__init__()
        self.finalConv = nn.Conv2d(in_channels=in_channels, out_channels=n_classes, kernel_size=kernel_size, stride=1,
                                   padding=kernel_size // 2, bias=True)
        self.softmax = nn.Softmax(dim=1)
        self.T = temperature

    def forward(self, x, useSoftmax=True):
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and self.finalConv.weight.requires_grad)):
            raise RuntimeError(
                "FCDenseNetClassifier.forward is inference-only on the HIP path: call it under torch.no_grad() / in eval "
                "mode, or differentiate through the owning module (TrainingBase.forward, FCDenseNet.forward, the fused "
                "training steps)")
        with torch.no_grad():
            return classifier_op(x, self.finalConv.weight, self.finalConv.bias, self.T, use_softmax=useSoftmax)


class FCDenseNet(nn.Module, EngineOwner):
    """tiramisu.py:128-147: featureExtractor + classifier in one module (fused forward)."""

    def __init__(self, in_channels=3, down_blocks=(5, 5, 5, 5, 5), up_blocks=(5, 5, 5, 5, 5), bottleneck_layers=5,
                 growth_rate=16, out_chans_first_conv=48, n_classes=12, kernel_size=1):
        super().__init__()
        self.featureExtractor = FCDenseNetFeatureExtractor(
            in_channels=in_channels, down_blocks=down_blocks, up_blocks=up_blocks,
            bottleneck_layers=bottleneck_layers, growth_rate=growth_rate, out_chans_first_conv=out_chans_first_conv)
        self.classifier = FCDenseNetClassifier(in_channels=self.featureExtractor.getFeatureChannels(),
                                               n_classes=n_classes, kernel_size=kernel_size)
        self.featureExtractor.__dict__["_rln_parent"] = self

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._rln_mark_dirty()
        return out

    def forward(self, x):
        wide = tuple(self.classifier.finalConv.kernel_size) != (1, 1)
        if self.training and torch.is_grad_enabled():  # differentiable (comparison.py-style user training code)
            if wide:
                raise RuntimeError("a classifier with kernel_size != 1 is inference-only on the HIP path (the fused "
                                   "backward covers the 1x1 classifier every reference script builds)")
            return ForwardFn.apply(self, x, None, None, *self._rln_params_in_arena_order())
        eng = self._rln_sync()
        with torch.no_grad():
            if wide:  # fused feature extractor, then the k x k classifier kernel (tiramisu.py:113-115)
                _, feat = eng.forward(x, training=self.training, with_backward=False, want_probs=False, want_feat=True)
                return classifier_op(feat, self.classifier.finalConv.weight, self.classifier.finalConv.bias,
                                     self.classifier.T)
            probs, _ = eng.forward(x, training=self.training, with_backward=False)
        return probs


# The named variants of the reference (tiramisu.py:150-194): layers per down block, per up block, bottleneck, growth.
# All of them start with a 48-channel first convolution on 3 input channels.
_VARIANTS = {
    "57": ((4,) * 5, (4,) * 5, 4, 12),
    "67": ((5,) * 5, (5,) * 5, 5, 16),
    "103": ((4, 5, 7, 10, 12), (12, 10, 7, 5, 4), 15, 16),
}


def _variant(name):
    down, up, bottleneck, growth = _VARIANTS[name]
    return dict(in_channels=3, down_blocks=down, up_blocks=up, bottleneck_layers=bottleneck, growth_rate=growth,
                out_chans_first_conv=48)


def FCDenseNet57(n_classes, kernel_size=1):
    return FCDenseNet(n_classes=n_classes, kernel_size=kernel_size, **_variant("57"))


def FCDenseNet67(n_classes):
    return FCDenseNet(n_classes=n_classes, **_variant("67"))


def FCDenseNet103(n_classes):
    return FCDenseNet(n_classes=n_classes, **_variant("103"))


def FCDenseNet57Base():
    return FCDenseNetFeatureExtractor(**_variant("57"))


def FCDenseNet57Classifier(n_classes):
    return FCDenseNetClassifier(FCDenseNet57Base().getFeatureChannels(), n_classes)


def FCDenseNet67Base():
    return FCDenseNetFeatureExtractor(**_variant("67"))


def FCDenseNet67Classifier(n_classes):
    return FCDenseNetClassifier(FCDenseNet67Base().getFeatureChannels(), n_classes)
