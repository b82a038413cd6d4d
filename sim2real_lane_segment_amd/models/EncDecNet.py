"""EncDecNet on the MI355X HIP path (API mirror of rightLaneNetwork/models/EncDecNet.py).

Legacy plain encoder-decoder (no reference script instantiates it; only its own __main__ block does).  Same class
names, constructor arguments, validation errors and state_dict keys (EncDecNet.py:5-116).  ``forward`` runs
Conv(k) -> activation -> BatchNorm2d -> Dropout -> MaxPool2d(k,2,k//2) / UpsamplingBilinear2d(x2) -> ... -> 1x1
conv -> Softmax(dim=-3) with HIP kernels from librln.so (igemm convolution with fused activation + statistics
epilogue; fused BN-affine/dropout/pool and BN-affine/dropout/bilinear kernels).  The reference defines no loss or
training step for this model, so this is a forward path (eval and train-mode statistics/dropout) without a backward.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib


class Conv(nn.Module):
    """conv -> activation -> batch norm -> dropout parameter container (EncDecNet.py:5-37)."""

    def __init__(self, inCh, oCh, kernelSize=3, stride=1, activation=nn.ReLU(), bNorm=True, dropOut=0.3):
        super().__init__()
        if dropOut >= 1 or dropOut < 0:
            raise ValueError(f"Conv dropOut must lie in [0,1); got {dropOut}")
        if stride != 1:
            raise ValueError("only stride 1 is used by EncDecNet and built on the HIP path")
        self.conv = nn.Conv2d(inCh, oCh, kernelSize, padding=kernelSize // 2, stride=stride)
        self.activation = activation
        self.bn = nn.BatchNorm2d(oCh) if bNorm else nn.Identity()
        self.drop = nn.Identity() if dropOut == 0 else nn.Dropout(p=dropOut)

    def forward(self, x):  # pragma: no cover - guard
        raise RuntimeError("Conv is a parameter container of the HIP path; call EncDecNet.forward")


activationDict = nn.ModuleDict({
    'relu': nn.ReLU(),
    'prelu': nn.PReLU(),
    'leakyRelu': nn.LeakyReLU(),
    'sigmoid': nn.Sigmoid(),
    'tanh': nn.Tanh(),
    'none': nn.Identity(),
})
activationTypes = list(activationDict.keys())


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _act_code(act):
    if isinstance(act, nn.ReLU):
        return 1, 0.0
    if isinstance(act, nn.PReLU):
        if act.weight.numel() != 1:
            raise RuntimeError("only the single-parameter PReLU of the reference is built")
        return 2, float(act.weight.detach().reshape(-1)[0])
    if isinstance(act, nn.LeakyReLU):
        return 2, float(act.negative_slope)
    if isinstance(act, nn.Sigmoid):
        return 3, 0.0
    if isinstance(act, nn.Tanh):
        return 4, 0.0
    return 0, 0.0


class EncDecNet(nn.Module):
    def __init__(self, nFeat: int, nLevels: int, kernelSize: int = 3, nLinType: str = 'relu', bNorm: bool = True,
                 dropOut: int = 0.3, inFeat=3):
        super().__init__()
        if nFeat < 1:
            raise ValueError(f"number of channels must be at least 1; got {nFeat}")
        if nLevels < 1:
            raise ValueError(f"number of levels must be at least 1; got {nLevels}")
        if nLinType not in activationTypes:
            raise ValueError(f"activation {nLinType} is not implemented; choose one of {activationTypes}")
        if kernelSize not in (1, 3, 7):
            raise ValueError("the HIP path builds kernel sizes 1, 3 and 7")
        self.kernelSize = kernelSize
        self.activation = activationDict[nLinType]
        self.pool = nn.MaxPool2d(kernelSize, stride=2, padding=kernelSize // 2)
        self.upsample = nn.UpsamplingBilinear2d(scale_factor=2)
        self.encoders = nn.ModuleList()
        self.decoders = nn.ModuleList()
        oFeat = nFeat
        for _ in range(nLevels):
            self.encoders.append(Conv(inFeat, oFeat, kernelSize=kernelSize, activation=self.activation, bNorm=bNorm,
                                      dropOut=dropOut))
            inFeat = oFeat
            oFeat = 2 * oFeat
        oFeat = oFeat // 2
        for _ in range(nLevels):
            self.decoders.append(Conv(inFeat, oFeat, kernelSize=kernelSize, activation=self.activation, bNorm=bNorm,
                                      dropOut=dropOut))
            inFeat = oFeat
            oFeat = oFeat // 2
        self.classifier = Conv(inFeat, 2, kernelSize=1, activation=nn.Softmax(dim=-3), bNorm=False, dropOut=0)
        self._seed = 0

    def getNParams(self):
        return sum(p.numel() for p in self.parameters())

    # ---- HIP forward --------------------------------------------------------------------------------------
    def _block(self, L, blk, x, mode, mask):
        """mode: 'pool' | 'up' | 'none'."""
        n, cin, h, w = x.shape
        cout = blk.conv.out_channels
        k = blk.conv.kernel_size[0]
        act, slope = _act_code(blk.activation)
        has_bn = isinstance(blk.bn, nn.BatchNorm2d)
        y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        stats = ws = None
        if has_bn and self.training:
            stats = torch.empty((cout, 2), dtype=torch.float32, device=x.device)
            ws = torch.empty(((n * ((h + 7) // 8) * ((w + 15) // 16) + 8) * cout * 2 * 4,), dtype=torch.uint8,
                             device=x.device)
        wt = blk.conv.weight.detach().contiguous()
        bs = blk.conv.bias.detach().contiguous()
        _lib.check(L.rln_op_conv_act(_p(x), n, cin, h, w, _p(wt), _p(bs), cout, k, act, slope, _p(y), _p(stats), _p(ws),
                                     ws.numel() if ws is not None else 0, _stream()), "rln_op_conv_act")
        a = b = None
        if has_bn:
            a = torch.empty(cout, dtype=torch.float32, device=x.device)
            b = torch.empty(cout, dtype=torch.float32, device=x.device)
            _lib.check(L.rln_op_bn_affine(_p(stats), cout, float(n * h * w), int(self.training), _p(blk.bn.weight),
                                          _p(blk.bn.bias), _p(blk.bn.running_mean), _p(blk.bn.running_var),
                                          float(blk.bn.momentum), float(blk.bn.eps), _p(a), _p(b), _stream()),
                       "rln_op_bn_affine")
            if self.training:
                blk.bn.num_batches_tracked += 1
        if mode == 'pool':
            ho, wo = (h + 2 * (self.kernelSize // 2) - self.kernelSize) // 2 + 1, \
                     (w + 2 * (self.kernelSize // 2) - self.kernelSize) // 2 + 1
            out = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
            _lib.check(L.rln_op_bn_drop_maxpool(_p(y), n, cout, h, w, _p(a), _p(b), _p(mask), self.kernelSize, _p(out),
                                                _stream()), "rln_op_bn_drop_maxpool")
            return out
        if mode == 'up':
            out = torch.empty((n, cout, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
            _lib.check(L.rln_op_bn_drop_upsample2(_p(y), n, cout, h, w, _p(a), _p(b), _p(mask), _p(out), _stream()),
                       "rln_op_bn_drop_upsample2")
            return out
        return y

    def _mask(self, L, blk, shape, device, injected):
        if not self.training or not isinstance(blk.drop, nn.Dropout):
            return None
        if injected is not None:
            return injected.to(device=device, dtype=torch.float32).contiguous()
        m = torch.empty(shape, dtype=torch.float32, device=device)
        self._seed += 1
        _lib.check(L.rln_op_dropout_mask(_p(m), m.numel(), 1.0 - blk.drop.p, self._seed, _stream()))
        return m

    def forward(self, x, drop_masks=None):
        """drop_masks: optional list (one per encoder/decoder, execution order) of [N,C,H,W] tensors holding 0 or
        1/(1-p) -- parity tests inject the oracle's masks; otherwise masks are drawn on the device."""
        if x.device.type != "cuda":
            raise RuntimeError("EncDecNet runs on the GPU only (HIP kernels); there is no CPU fallback")
        L = _lib.lib()
        x = x.float().contiguous()
        with torch.no_grad():
            i = 0
            for blk in self.encoders:
                shape = (x.shape[0], blk.conv.out_channels, x.shape[2], x.shape[3])
                m = self._mask(L, blk, shape, x.device, drop_masks[i] if drop_masks is not None else None)
                x = self._block(L, blk, x, 'pool', m)
                i += 1
            for blk in self.decoders:
                shape = (x.shape[0], blk.conv.out_channels, x.shape[2], x.shape[3])
                m = self._mask(L, blk, shape, x.device, drop_masks[i] if drop_masks is not None else None)
                x = self._block(L, blk, x, 'up', m)
                i += 1
            logits = self._block(L, self.classifier, x, 'none', None)
            n, c, h, w = logits.shape
            out = torch.empty_like(logits)
            _lib.check(L.rln_op_softmax_channels(_p(logits), n, c, h * w, _p(out), _stream()))
        return out
