"""Parameter containers mirroring rightLaneNetwork/models/FCDenseNet/layers.py.

Same class names, constructor signatures, sub-module names and therefore the same state_dict
keys as the reference (layers.py:5-12,18-24,43-52,58-63,72-76).  They hold parameters only: the
arithmetic of a whole feature extractor runs in librln.so through the owning module
(``FCDenseNetFeatureExtractor.forward``), which lays these parameters out in one flat device
arena.  A sub-block is not a separately executable unit on this path, so calling one directly
raises instead of silently running stock PyTorch operators.
"""
import torch.nn as nn


class _HostOnly(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container of the HIP path; run the enclosing "
            f"FCDenseNetFeatureExtractor / TrainingBase module instead (no PyTorch-operator fallback).")


class DenseLayer(_HostOnly):
    """BN -> ReLU -> Conv3x3(in, growth) -> Dropout2d(0.2)   (layers.py:5-12)."""

    def __init__(self, in_channels, growth_rate):
        super().__init__()
        self.norm = nn.BatchNorm2d(in_channels)
        self.relu = nn.ReLU(True)
        self.conv = nn.Conv2d(in_channels, growth_rate, kernel_size=3, stride=1, padding=1, bias=True)
        self.drop = nn.Dropout2d(0.2)


class DenseBlock(_HostOnly):
    """n_layers DenseLayers on a growing channel stack (layers.py:18-40)."""

    def __init__(self, in_channels, growth_rate, n_layers, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.layers = nn.ModuleList(
            [DenseLayer(in_channels + i * growth_rate, growth_rate) for i in range(n_layers)])


class TransitionDown(_HostOnly):
    """BN -> ReLU -> Conv1x1 -> Dropout2d(0.2) -> MaxPool2d(2)   (layers.py:43-52)."""

    def __init__(self, in_channels):
        super().__init__()
        self.norm = nn.BatchNorm2d(num_features=in_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0, bias=True)
        self.drop = nn.Dropout2d(0.2)
        self.maxpool = nn.MaxPool2d(2)


class TransitionUp(_HostOnly):
    """ConvTranspose2d(k3, s2, p0) -> crop to the skip -> concat   (layers.py:58-69)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.convTrans = nn.ConvTranspose2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3,
                                            stride=2, padding=0, bias=True)


class Bottleneck(_HostOnly):
    """DenseBlock(upsample=True) wrapped under the name 'bottleneck' (layers.py:72-76)."""

    def __init__(self, in_channels, growth_rate, n_layers):
        super().__init__()
        self.bottleneck = DenseBlock(in_channels, growth_rate, n_layers, upsample=True)


def center_crop(layer, max_height, max_width):
    """layers.py:82-86 (pure view arithmetic; the HIP transposed-conv kernel writes only this region)."""
    _, _, h, w = layer.size()
    x0 = (w - max_width) // 2
    y0 = (h - max_height) // 2
    return layer[:, :, y0:y0 + max_height, x0:x0 + max_width]
