"""Host-side owner of one ``rln_ctx``: flat device arenas, workspace, and the forward / loss /
backward / AdamW calls of the C ABI (include/rln.h).  PyTorch is used for device memory and
streams only; every arithmetic op of the path runs in librln.so.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class NetSpec:
    """Constructor knobs of FCDenseNetFeatureExtractor/Classifier (models/FCDenseNet/tiramisu.py:21-24,112-118)."""
    in_channels: int = 3
    down_blocks: Tuple[int, ...] = (5, 5, 5, 5, 5)
    up_blocks: Tuple[int, ...] = (5, 5, 5, 5, 5)
    bottleneck_layers: int = 5
    growth_rate: int = 16
    out_chans_first_conv: int = 48
    n_classes: int = 2
    temperature: float = 0.05


@dataclass
class TensorMeta:
    name: str
    kind: int
    offset: int
    shape: Tuple[int, ...]

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def parse_dense_arith(text):
    """'f16x2,bf16x2' -> (2, 'f16', 2, 'bf16'); 'fp32' = exact fp32 MFMA kernels (0 parts)."""
    def one(a):
        a = a.strip()
        if a in ("", "fp32"):
            return 0, "bf16"
        t, n = a.split("x")
        return int(n), t
    f, _, b = text.partition(",")
    fp, ft = one(f)
    bp, bt = one(b)
    return fp, ft, bp, bt


# Arithmetic of the dense 3x3 layers used by every new Engine unless the caller passes dense_arith= (include/rln.h:
# rln_set_dense_arith): forward on f16 operands split into 2 parts (3 products, ~2^-22: the accuracy class of the exact
# fp32 MFMA chain, tools/dense3_precision.py), backward on bf16 x 2 parts (gradients need bf16's exponent range).
# Storage and accumulation are fp32 in every mode.  "fp32,fp32" selects the exact-fp32 MFMA kernels of round 1.
# RLN_DENSE_ARITH overrides the default for experiments.
DEFAULT_DENSE_ARITH = parse_dense_arith(os.environ.get("RLN_DENSE_ARITH") or "f16x2,bf16x2")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One network instance on one device.

    ``params``/``grads`` are flat fp32 arenas in forward execution order; ``views[name]`` are
    tensors aliasing them under the reference's state_dict names.
    """

    def __init__(self, spec: NetSpec, device="cpu", dense_arith=None):
        self.spec = spec
        self.L = _lib.lib()
        cfg = _lib.make_config(spec.in_channels, spec.down_blocks, spec.up_blocks, spec.bottleneck_layers,
                               spec.growth_rate, spec.out_chans_first_conv, spec.n_classes, spec.temperature)
        handle = ctypes.c_void_p()
        _lib.check(self.L.rln_create(ctypes.byref(cfg), ctypes.byref(handle)), "rln_create")
        self.ctx = handle
        self.metas: List[TensorMeta] = []
        name = ctypes.create_string_buffer(256)
        kind, off, ndim = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int()
        shape = (ctypes.c_int64 * 4)()
        for i in range(self.L.rln_num_tensors(self.ctx)):
            _lib.check(self.L.rln_tensor_info(self.ctx, i, name, 256, ctypes.byref(kind), ctypes.byref(off),
                                              ctypes.byref(ndim), shape), "rln_tensor_info")
            self.metas.append(TensorMeta(name.value.decode(), kind.value, off.value,
                                         tuple(int(shape[k]) for k in range(ndim.value))))
        self.n_param = int(self.L.rln_param_count(self.ctx))
        self.n_bnstat = int(self.L.rln_bnstat_count(self.ctx))
        self.n_nbt = int(self.L.rln_nbt_count(self.ctx))
        self.feature_channels = int(self.L.rln_feature_channels(self.ctx))
        nd = self.L.rln_num_dropouts(self.ctx)
        per = (ctypes.c_int * max(nd, 1))()
        self.drop_total = int(self.L.rln_dropout_channels(self.ctx, per))
        self.drop_channels = [int(per[i]) for i in range(nd)]
        self.n_seg = int(self.L.rln_backward_segments(self.ctx))
        self.seg_ranges = []
        b, e = ctypes.c_int64(), ctypes.c_int64()
        for s in range(self.n_seg):
            _lib.check(self.L.rln_backward_segment_range(self.ctx, s, ctypes.byref(b), ctypes.byref(e)))
            self.seg_ranges.append((b.value, e.value))
        self.device = torch.device("cpu")
        self.params = self.grads = self.bnrun = self.nbt = None
        self._eval_cache = False
        self._eval_key = None
        self._param_epoch = 0  # bumped by the engine-side optimiser steps (they write the arena through raw pointers)
        self.views: Dict[str, torch.Tensor] = {}
        self._grad_views: Dict[str, torch.Tensor] = {}
        self._grad_views_of = None
        self._ws = None
        self._ws_key = None
        self._keep = []
        self.step_seed = 0
        self.fwd_token = 0
        self.dense_arith = (0, "bf16", 0, "bf16")
        self.storage = "f32"
        self.allocate(device)
        arith = DEFAULT_DENSE_ARITH if dense_arith is None else dense_arith
        if arith is not None:
            self.set_dense_arith(*arith)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.L.rln_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ---- arenas ---------------------------------------------------------------------------
    def allocate(self, device, old: Optional["Engine"] = None):
        """(Re)allocates the arenas on ``device`` keeping current values."""
        device = torch.device(device)
        prev = (self.params, self.bnrun, self.nbt)
        self.params = torch.zeros(self.n_param, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.n_param, dtype=torch.float32, device=device)
        self.bnrun = torch.zeros(self.n_bnstat, dtype=torch.float32, device=device)
        self.nbt = torch.zeros(self.n_nbt, dtype=torch.int64, device=device)
        if prev[0] is not None:
            self.params.copy_(prev[0])
            self.bnrun.copy_(prev[1])
            self.nbt.copy_(prev[2])
        self.device = device
        self.views = {}
        self._grad_views_of = None
        for m in self.metas:
            if m.kind == _lib.T_PARAM:
                self.views[m.name] = self.params[m.offset:m.offset + m.numel].view(m.shape)
            elif m.kind in (_lib.T_RUNNING_MEAN, _lib.T_RUNNING_VAR):
                self.views[m.name] = self.bnrun[m.offset:m.offset + m.numel].view(m.shape)
            else:
                self.views[m.name] = self.nbt[m.offset]
        self._ws = None
        self._ws_key = None
        if device.type == "cuda":
            _lib.check(self.L.rln_bind_params(self.ctx, _ptr(self.params), _ptr(self.grads), _ptr(self.bnrun),
                                              _ptr(self.nbt)), "rln_bind_params")

    @property
    def grad_views(self) -> Dict[str, torch.Tensor]:
        """Views of the engine's own gradient arena under the reference's parameter names."""
        if self._grad_views_of is not self.grads:
            self._grad_views = {m.name: self.grads[m.offset:m.offset + m.numel].view(m.shape)
                                for m in self.metas if m.kind == _lib.T_PARAM}
            self._grad_views_of = self.grads
        return self._grad_views

    def bind_grads(self, flat: Optional[torch.Tensor]):
        """Makes ``flat`` (fp32, n_param elements, on the engine's device) the gradient arena of the backward calls
        that follow (rln_bind_grads); None re-binds the engine's own arena ``self.grads``.  The module path binds a
        fresh buffer around every backward (autograd then owns it) and re-binds the own arena right after."""
        self._require_gpu()
        if flat is None:
            flat = self.grads
        if flat.dtype != torch.float32 or flat.numel() != self.n_param or flat.device != self.device or \
                not flat.is_contiguous():
            raise RuntimeError("gradient arena must be a contiguous fp32 tensor of n_param elements on the engine's device")
        _lib.check(self.L.rln_bind_grads(self.ctx, _ptr(flat)), "rln_bind_grads")

    def load_state(self, state: Dict[str, torch.Tensor]):
        for m in self.metas:
            self.views[m.name].copy_(state[m.name])

    def state(self) -> Dict[str, torch.Tensor]:
        return {m.name: self.views[m.name].detach().clone() for m in self.metas}

    # ---- arithmetic of the dense 3x3 layers -------------------------------------------------
    DTYPES = {"bf16": 0, "f16": 1}

    def set_dense_arith(self, fwd_parts=0, fwd_dtype="bf16", bwd_parts=0, bwd_dtype="bf16"):
        """0 parts = exact fp32 MFMA kernels; 1..3 = 16-bit MFMA on operands split into that many parts
        (rln_set_dense_arith, include/rln.h).  Storage stays fp32 in every mode."""
        _lib.check(self.L.rln_set_dense_arith(self.ctx, int(fwd_parts), self.DTYPES[fwd_dtype], int(bwd_parts),
                                              self.DTYPES[bwd_dtype]), "rln_set_dense_arith")
        self.dense_arith = (int(fwd_parts), fwd_dtype, int(bwd_parts), bwd_dtype)
        self._ws = None
        self._ws_key = None

    def set_storage(self, mode):
        """'f32' (default) or 'bf16': element type of the activation stacks / finalised output gradients in HBM
        (rln_set_storage, include/rln.h).  'bf16' switches the arithmetic to one-part bf16 operands."""
        code = {"f32": 0, "fp32": 0, "bf16": 1, 0: 0, 1: 1}[mode]
        _lib.check(self.L.rln_set_storage(self.ctx, code), "rln_set_storage")
        self.storage = "bf16" if code else "f32"
        if code:
            self.dense_arith = (1, "bf16", 1, "bf16")
        self._ws = None
        self._ws_key = None

    def set_eval_cache(self, enable=True):
        """Frozen-model loops (makeDemoVideo.py:15-47, test.py:80-94): eval forwards reuse the weight fragments and the
        folded BatchNorm tables of the previous eval forward instead of rebuilding them on every call
        (rln_set_eval_cache, include/rln.h).  The tables are rebuilt whenever torch's version counters of the parameter
        / running-statistics arenas moved (optimiser steps, load_state_dict, any in-place write through the module's
        parameters) or an engine-side optimiser step ran; writes that bypass the counters (`tensor.data`, raw
        pointers) need another set_eval_cache(True) call.  Results are bit-identical to the uncached forward."""
        self._eval_cache = bool(enable)
        self._eval_key = None
        _lib.check(self.L.rln_set_eval_cache(self.ctx, int(self._eval_cache)), "rln_set_eval_cache")

    def _eval_cache_check(self):
        key = (self.params._version, self.bnrun._version, self._param_epoch, self.params.data_ptr())
        if key != self._eval_key:
            _lib.check(self.L.rln_set_eval_cache(self.ctx, 1), "rln_set_eval_cache")  # invalidates
            self._eval_key = key

    @property
    def wgrad_parts(self):
        return int(self.L.rln_get_wgrad_parts(self.ctx))

    def set_wgrad_parts(self, parts):
        """Operand parts of the dense weight-gradient GEMMs (rln_set_wgrad_parts; 0 = as the backward arithmetic)."""
        _lib.check(self.L.rln_set_wgrad_parts(self.ctx, int(parts)), "rln_set_wgrad_parts")

    # ---- workspace ------------------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("the lane-segmentation path runs on the GPU only (HIP kernels); "
                               "move the module with .cuda() -- there is no CPU fallback")

    def ensure_workspace(self, n, h, w, with_backward):
        self._require_gpu()
        key = (n, h, w)
        if self._ws_key is not None and self._ws_key[:3] == key and (self._ws_key[3] or not with_backward):
            return
        need = int(self.L.rln_workspace_bytes(self.ctx, n, h, w, int(with_backward)))
        if self._ws is not None:
            # kernels of the previous geometry may still be running on the block that is about to be handed back to the
            # allocator and carved again (rln_set_workspace writes descriptor tables into it from the host)
            torch.cuda.synchronize(self.device)
        self._ws = None  # release before allocating the next one
        guard = int(os.environ.get("RLN_WS_GUARD", "0"))  # debugging aid: extra tail bytes, checked by check_ws_guard()
        self._ws = torch.empty(need + 256 + guard, dtype=torch.uint8, device=self.device)
        self._ws_need = need
        if guard:
            self._ws[-guard:].fill_(0xA5)
        if os.environ.get("RLN_POISON_WORKSPACE"):
            # debugging aid (tools/poison_probe.py): every byte 0xFF = NaN in fp32 and bf16.  A kernel that reads workspace
            # it has not written in this step shows up as non-finite / changed results instead of depending on what the
            # allocator's block happened to hold.
            self._ws[:need + 256].fill_(0xFF)
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        _lib.check(self.L.rln_set_workspace(self.ctx, ctypes.c_void_p(aligned), need, n, h, w, int(with_backward)),
                   "rln_set_workspace")
        self._ws_key = (n, h, w, bool(with_backward))

    def check_ws_guard(self):
        """True when the RLN_WS_GUARD tail behind the workspace is untouched (no kernel wrote past the carved size)."""
        guard = int(os.environ.get("RLN_WS_GUARD", "0"))
        if not guard or self._ws is None:
            return True
        torch.cuda.synchronize()
        return bool((self._ws[-guard:] == 0xA5).all())

    # ---- compute --------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, training: bool, with_backward: bool = False,
                drop_scales: Optional[torch.Tensor] = None, seed: Optional[int] = None, want_probs=True,
                want_feat=False, use_softmax=True):
        self._require_gpu()
        if x.dim() != 4 or x.shape[1] != self.spec.in_channels:
            raise RuntimeError(f"expected input [N,{self.spec.in_channels},H,W], got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        n, _, h, w = x.shape
        self.ensure_workspace(n, h, w, with_backward)
        probs = torch.empty((n, self.spec.n_classes, h, w), dtype=torch.float32, device=self.device) \
            if want_probs else None
        feat = torch.empty((n, self.feature_channels, h, w), dtype=torch.float32, device=self.device) \
            if want_feat else None
        if drop_scales is not None:
            drop_scales = drop_scales.to(device=self.device, dtype=torch.float32).contiguous()
            if drop_scales.numel() != n * self.drop_total:
                raise RuntimeError("drop_scales has the wrong size")
        if seed is None:
            self.step_seed += 1
            seed = self.step_seed
        if self._eval_cache and not training:
            self._eval_cache_check()
        _lib.check(self.L.rln_forward(self.ctx, _ptr(x), n, h, w, int(training), _ptr(drop_scales), int(seed),
                                      _ptr(probs), _ptr(feat), int(use_softmax), _stream()), "rln_forward")
        self._keep = [x, drop_scales]  # the backward pass re-reads the input
        self.fwd_token += 1            # a backward belongs to exactly one forward (see TrainStepFn / ForwardFn)
        return probs, feat

    def pack_drop_scales(self, scales: Sequence[torch.Tensor]) -> torch.Tensor:
        """list of [N, C_call] tensors (oracle layout) -> flat call-major device tensor."""
        assert len(scales) == len(self.drop_channels)
        return torch.cat([s.reshape(-1).float() for s in scales]).to(self.device)

    def loss(self, probs: torch.Tensor, y: torch.Tensor, weighted: bool, want_argmax=False, want_confusion=False):
        self._require_gpu()
        n, k, h, w = probs.shape
        y = y.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(3 + k, dtype=torch.float32, device=self.device)
        am = torch.empty((n, h, w), dtype=torch.int64, device=self.device) if want_argmax else None
        conf = torch.empty((k, k), dtype=torch.int64, device=self.device) if want_confusion else None
        _lib.check(self.L.rln_loss(self.ctx, _ptr(probs), _ptr(y), n, h, w, int(weighted), _ptr(out), _ptr(am),
                                   _ptr(conf), _stream()), "rln_loss")
        self._keep.append(y)
        return out, am, conf

    def entropy_loss(self, probs: torch.Tensor, lamda: float):
        """adentropy of the MME unlabelled branch; a following backward() applies the gradient reversal."""
        self._require_gpu()
        n, k, h, w = probs.shape
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.L.rln_entropy_loss(self.ctx, _ptr(probs), n, h, w, float(lamda), _ptr(out), _stream()),
                   "rln_entropy_loss")
        return out

    def set_output_grad(self, dprobs: torch.Tensor):
        """d(loss)/d(probabilities) of the last training forward; a following backward() differentiates through the
        whole net (rln_set_output_grad)."""
        self._require_gpu()
        dprobs = dprobs.to(device=self.device, dtype=torch.float32).contiguous()
        n, k, h, w = dprobs.shape
        _lib.check(self.L.rln_set_output_grad(self.ctx, _ptr(dprobs), n, h, w), "rln_set_output_grad")
        self._keep.append(dprobs)

    def sgd_step(self, momentum_buf, lo, hi, lr, momentum, weight_decay, first_step, grads_ptr=None, grad_scale=1.0):
        """Nesterov SGD on the arena range [lo, hi) (one parameter group)."""
        self._require_gpu()
        gptr = ctypes.c_void_p((grads_ptr if grads_ptr is not None else self.grads.data_ptr()) + 4 * lo)
        self._param_epoch += 1
        _lib.check(self.L.rln_sgd_step(_ptr(self.params[lo:hi]), gptr, _ptr(momentum_buf[lo:hi]), hi - lo, float(lr),
                                       float(momentum), float(weight_decay), int(first_step), float(grad_scale),
                                       _stream()), "rln_sgd_step")

    def backward(self, loss_scale: float = 1.0, seg_begin: int = 0, seg_end: Optional[int] = None,
                 loss_scale_dev: Optional[torch.Tensor] = None):
        """loss_scale_dev: optional device scalar multiplied on top of loss_scale inside the head-backward kernel
        (rln_backward_scaled): the d(loss) autograd hands to loss.backward() never visits the host."""
        self._require_gpu()
        if seg_end is None:
            seg_end = self.n_seg
        if loss_scale_dev is not None:
            if loss_scale_dev.dtype != torch.float32 or loss_scale_dev.device != self.device:
                loss_scale_dev = loss_scale_dev.to(device=self.device, dtype=torch.float32)
            self._keep.append(loss_scale_dev)
        _lib.check(self.L.rln_backward_scaled(self.ctx, float(loss_scale), _ptr(loss_scale_dev), seg_begin, seg_end,
                                              _stream()), "rln_backward_scaled")

    def adamw_step(self, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                   grad_scale=1.0, lo=0, hi=None):
        self._require_gpu()
        hi = self.n_param if hi is None else hi
        self._param_epoch += 1
        _lib.check(self.L.rln_adamw_step(_ptr(self.params[lo:hi]), _ptr(self.grads[lo:hi]), _ptr(exp_avg[lo:hi]),
                                         _ptr(exp_avg_sq[lo:hi]), hi - lo, lr, betas[0], betas[1], eps,
                                         weight_decay, int(step), grad_scale, _stream()), "rln_adamw_step")


def _adamw_step_ptr(self, grads_ptr, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                    grad_scale=1.0):
    """AdamW over the whole arena with gradients at an arbitrary flat device address."""
    self._require_gpu()
    self._param_epoch += 1
    _lib.check(self.L.rln_adamw_step(_ptr(self.params), ctypes.c_void_p(grads_ptr), _ptr(exp_avg), _ptr(exp_avg_sq),
                                     self.n_param, lr, betas[0], betas[1], eps, weight_decay, int(step), grad_scale,
                                     _stream()), "rln_adamw_step")


Engine.adamw_step_ptr = _adamw_step_ptr


def classifier_op(feat, weight, bias, temperature, use_softmax=True):
    """FCDenseNetClassifier.forward on arbitrary features (tiramisu.py:120-125)."""
    L = _lib.lib()
    if feat.device.type != "cuda":
        raise RuntimeError("classifier runs on the GPU only (HIP kernels); no CPU fallback")
    feat = feat.float().contiguous()
    n, c, h, w = feat.shape
    k = weight.shape[0]
    ks = int(weight.shape[-1]) if weight.dim() == 4 else 1
    if ks != 1:
        # finalConv with kernel_size != 1 (tiramisu.py:113-115; FCDenseNet57(n_classes, kernel_size)): the k x k implicit
        # GEMM with padding k // 2, then / T and the channel softmax
        if weight.shape[-1] != weight.shape[-2] or ks not in (3, 7):
            raise RuntimeError(f"classifier kernel_size {tuple(weight.shape[-2:])}: 1, 3 and 7 are built on the HIP path")
        wt = weight.detach().float().contiguous()
        bt = bias.detach().float().contiguous()
        logits = torch.empty((n, k, h, w), dtype=torch.float32, device=feat.device)
        _lib.check(L.rln_op_conv_act(_ptr(feat), n, c, h, w, _ptr(wt), _ptr(bt), k, ks, 0, 0.0, _ptr(logits), None, None, 0,
                                     _stream()), "rln_op_conv_act")
        out = torch.empty_like(logits)
        _lib.check(L.rln_op_scaled_softmax(_ptr(logits), n, k, h * w, float(temperature), int(use_softmax), _ptr(out),
                                           _stream()), "rln_op_scaled_softmax")
        return out
    wt = weight.detach().reshape(k, c).float().contiguous()
    bt = bias.detach().float().contiguous()
    out = torch.empty((n, k, h, w), dtype=torch.float32, device=feat.device)
    _lib.check(L.rln_op_classifier(_ptr(feat), n, c, h * w, _ptr(wt), _ptr(bt), k, float(temperature), _ptr(out),
                                   int(use_softmax), _stream()), "rln_op_classifier")
    return out
