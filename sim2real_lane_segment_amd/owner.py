"""Glue between torch.nn parameter containers and the flat-arena HIP engine.

An *owner* is a module holding a ``featureExtractor`` (and optionally a ``classifier``): it creates
one ``Engine`` and re-points every Parameter / buffer of those sub-modules at views of the
engine's flat device arenas (``param.data = view`` keeps Parameter identity, so optimisers,
``state_dict``/``load_state_dict``, ``.parameters()`` and Lightning keep working unchanged).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib
from .engine import Engine, NetSpec


def _resolve(root: nn.Module, dotted: str):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
    return mod, parts[-1]


class EngineOwner:
    """Mixin. Subclass must be an nn.Module with ``featureExtractor`` (+ optional ``classifier``)."""

    _rln_engine: Optional[Engine] = None
    _rln_dirty: bool = True
    _rln_dummy_classifier = None
    _rln_reducer = None

    def rln_eval_cache(self, enable: bool = True):
        """Frozen-model loops (makeDemoVideo.py:15-47, test.py:80-94: `model.eval()`, one forward per frame): eval
        forwards reuse the MFMA weight fragments and the folded BatchNorm tables instead of rebuilding them per call
        (Engine.set_eval_cache).  In-place writes through the module's parameters / buffers (optimisers,
        load_state_dict) are noticed through torch's version counters; `tensor.data` writes are not -- call this
        again after those.  Off by default."""
        self._rln_sync().set_eval_cache(enable)
        return self

    def enable_grad_allreduce(self, n_buckets: int = 4, group=None, force_collectives: bool = False):
        """Data-parallel training through the module path (what Lightning drives: training_step -> loss.backward() ->
        optimizer.step(), train.py:63-64): every backward of this module runs its segments bucket by bucket and
        all-reduces each finished slice of the flat gradient buffer on a side stream while the remaining backward
        runs (trainer.BucketedGradReducer); autograd receives the MEAN over ranks, as torch DDP would deliver it.
        Call once after torch.distributed.init_process_group; do not wrap the module in DistributedDataParallel as
        well (the gradients would be reduced twice)."""
        from .trainer import BucketedGradReducer
        eng = self._rln_sync()
        object.__setattr__(self, "_rln_reducer",
                           BucketedGradReducer(eng.grads, eng.seg_ranges, n_buckets, group=group,
                                               force_collectives=force_collectives))
        return self._rln_reducer

    def _rln_backward_into_fresh_arena(self, eng, g_scale):
        """One backward into a fresh flat buffer that autograd then owns (AccumulateGrad may keep or add into the
        views): no copy out of a fixed arena, d(loss) applied inside the head-backward kernel from its device scalar."""
        flat = torch.empty_like(eng.params)
        eng.bind_grads(flat)
        try:
            red = self._rln_reducer
            if red is not None and (red.world > 1 or red.force):
                inv = 1.0 / red.world  # the all-reduce sums: scaling d(loss) by 1/world delivers the mean
                red.backward_and_reduce(lambda sb, se: eng.backward(inv, sb, se, loss_scale_dev=g_scale), flat=flat)
            else:
                eng.backward(1.0, loss_scale_dev=g_scale)
        finally:
            eng.bind_grads(None)  # kernel arguments are baked at enqueue time: the own arena is current again
        return tuple(flat[m.offset:m.offset + m.numel].view(m.shape) for m in eng.metas
                     if m.kind == _lib.T_PARAM and m.name in self._rln_param_set)

    def _rln_spec(self) -> NetSpec:
        fe = self.featureExtractor
        cl = getattr(self, "classifier", None)
        n_cls = cl.finalConv.out_channels if cl is not None else 1
        temp = cl.T if cl is not None else 0.05
        # a classifier whose finalConv is not 1x1 (FCDenseNet57(n_classes, kernel_size)) runs as its own k x k kernel behind
        # the fused feature extractor (inference); the fused net then carries a 1x1 stand-in that is never read
        self.__dict__["_rln_wide_classifier"] = cl is not None and tuple(cl.finalConv.kernel_size) != (1, 1)
        return NetSpec(in_channels=fe.in_channels, down_blocks=tuple(fe.down_blocks), up_blocks=tuple(fe.up_blocks),
                       bottleneck_layers=fe.bottleneck_layers, growth_rate=fe.growth_rate,
                       out_chans_first_conv=fe.out_chans_first_conv, n_classes=n_cls, temperature=float(temp))

    def _rln_named_tensors(self) -> Dict[str, torch.Tensor]:
        out = {}
        for prefix in ("featureExtractor", "classifier"):
            mod = getattr(self, prefix, None)
            if mod is None:
                continue
            for k, v in mod.state_dict(keep_vars=True).items():
                out[f"{prefix}.{k}"] = v
        return out

    def _rln_current_device(self):
        return next(self.featureExtractor.parameters()).device

    def _rln_mark_dirty(self):
        object.__setattr__(self, "_rln_dirty", True)

    def _rln_sync(self) -> Engine:
        """Makes sure every parameter/buffer aliases the engine arenas on the module's device."""
        dev = self._rln_current_device()
        eng = self._rln_engine
        if eng is not None and not self._rln_dirty and eng.device == dev:
            return eng
        current = self._rln_named_tensors()
        if eng is None:
            eng = Engine(self._rln_spec(), device=dev)
            object.__setattr__(self, "_rln_engine", eng)
        else:
            eng.allocate(dev)
        names = []
        for m in eng.metas:
            src = current.get(m.name)
            if self.__dict__.get("_rln_wide_classifier") and m.name.startswith("classifier."):
                continue  # the k x k classifier keeps its own tensors (its kernel reads them directly)
            if src is None:
                if m.name.startswith("classifier."):  # feature extractor used alone: dummy classifier stays zero
                    continue
                raise RuntimeError(f"module has no tensor named {m.name}")
            view = eng.views[m.name]
            with torch.no_grad():
                view.copy_(src.detach().to(view.device))
            root_name, rest = m.name.split(".", 1)
            holder, attr = _resolve(getattr(self, root_name), rest)
            if attr in holder._parameters:
                holder._parameters[attr].data = view
                holder._parameters[attr].grad = None
                names.append(m.name)
            else:
                holder._buffers[attr] = view
        object.__setattr__(self, "_rln_param_names", names)
        object.__setattr__(self, "_rln_param_set", set(names))
        object.__setattr__(self, "_rln_param_offsets", {m.name: m.offset for m in eng.metas if m.kind == _lib.T_PARAM})
        object.__setattr__(self, "_rln_dirty", False)
        return eng

    def _rln_params_in_arena_order(self) -> List[nn.Parameter]:
        eng = self._rln_sync()
        out = []
        for name in self._rln_param_names:
            root_name, rest = name.split(".", 1)
            holder, attr = _resolve(getattr(self, root_name), rest)
            out.append(holder._parameters[attr])
        return out


class TrainStepFn(torch.autograd.Function):
    """loss = weighted CE(softmax probs) of the whole net; backward fills the flat gradient arena."""

    @staticmethod
    def forward(ctx, owner, x, y, drop_scales, seed, *params):
        """y: int64 labels -> class-weighted CE step; y: float lamda -> MME unlabelled step (entropy of the
        classifier output behind a gradient-reversal layer, MMETrainingModule.py:28-33)."""
        if x.requires_grad:
            raise RuntimeError("the fused training step does not produce a gradient with respect to its input image "
                               "(x.requires_grad is set): detach the input")
        eng = owner._rln_sync()
        probs, _ = eng.forward(x, training=True, with_backward=True, drop_scales=drop_scales, seed=seed)
        if isinstance(y, float):
            out = eng.entropy_loss(probs, y)
        else:
            out, _, _ = eng.loss(probs, y, weighted=True)
        ctx.owner = owner
        ctx.n_params = len(params)
        ctx.token = eng.fwd_token
        loss = out[0].clone()
        extra = out.detach()
        ctx.mark_non_differentiable(extra, probs)
        return loss, extra, probs

    @staticmethod
    def backward(ctx, g_loss, g_extra, g_probs):
        owner = ctx.owner
        eng = owner._rln_engine
        _check_token(eng, ctx.token)
        # autograd receives slices of ONE fresh flat buffer (FusedAdamW recognises the layout and consumes it with one
        # kernel); with enable_grad_allreduce() the slices are all-reduced while the rest of the backward runs
        return (None, None, None, None, None) + owner._rln_backward_into_fresh_arena(eng, g_loss)


def _check_token(eng, token):
    """The engine keeps the activations of its LAST forward only: a backward of an older forward would silently use the
    wrong ones, so it fails loudly instead."""
    if eng.fwd_token != token:
        raise RuntimeError("backward of a stale forward: the engine keeps the activations of the most recent forward "
                           "only (another forward ran in between); run backward before the next forward, or use "
                           "separate modules")


class ForwardFn(torch.autograd.Function):
    """Differentiable ``forward`` of the fused net (train mode): probabilities out, d(loss)/d(probabilities) in
    (TrainingBase.forward used inside user-written steps: SimpleTrain.py:15, MMETrainingModule.py:34-35).  The gradient
    with respect to the input image is not produced (the reference's data loaders never ask for it)."""

    @staticmethod
    def forward(ctx, owner, x, drop_scales, seed, *params):
        if x.requires_grad:
            raise RuntimeError("the fused forward does not produce a gradient with respect to its input image "
                               "(x.requires_grad is set): detach the input")
        eng = owner._rln_sync()
        probs, _ = eng.forward(x, training=True, with_backward=True, drop_scales=drop_scales, seed=seed)
        ctx.owner = owner
        ctx.token = eng.fwd_token
        return probs

    @staticmethod
    def backward(ctx, g_probs):
        owner = ctx.owner
        eng = owner._rln_engine
        _check_token(eng, ctx.token)
        eng.set_output_grad(g_probs)
        return (None, None, None, None) + owner._rln_backward_into_fresh_arena(eng, None)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (single group, as SimpleTrain.py:28 configures it) executed by
    one HIP kernel over the flat parameter / gradient / moment arenas."""

    def __init__(self, owner: EngineOwner, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.owner = owner
        params = owner._rln_params_in_arena_order()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        eng = owner._rln_engine
        self._step = 0
        self.exp_avg = torch.zeros_like(eng.params)
        self.exp_avg_sq = torch.zeros_like(eng.params)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self.owner._rln_sync()
        if self.exp_avg.device != eng.params.device:
            self.exp_avg = self.exp_avg.to(eng.params.device)
            self.exp_avg_sq = self.exp_avg_sq.to(eng.params.device)
        group = self.param_groups[0]
        names = self.owner._rln_param_names
        offs = self.owner._rln_param_offsets
        params = group["params"]
        # fast path: every .grad is a slice of ONE flat buffer in arena order (what TrainStepFn returns)
        flat_ptr = None
        g0 = params[0].grad
        if g0 is not None and g0.dtype == torch.float32 and g0.device == eng.params.device:
            base = g0.data_ptr() - 4 * offs[names[0]]
            if all(p.grad is not None and p.grad.data_ptr() == base + 4 * offs[n] for p, n in zip(params, names)):
                flat_ptr = base
        if flat_ptr is None:
            for p, name in zip(params, names):  # gradients from another autograd path: bring them into the arena
                gv = eng.grad_views[name]
                if p.grad is None:
                    gv.zero_()
                elif p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
            flat_ptr = eng.grads.data_ptr()
        self._step += 1
        eng.adamw_step_ptr(flat_ptr, self.exp_avg, self.exp_avg_sq, self._step, float(group["lr"]),
                           tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"]))
        return loss

    def state_dict(self):
        d = super().state_dict()
        d["rln"] = {"step": self._step, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
        return d

    def load_state_dict(self, state_dict):
        extra = state_dict.get("rln")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "rln"})
        if extra is not None:
            self._step = int(extra["step"])
            self.exp_avg.copy_(extra["exp_avg"])
            self.exp_avg_sq.copy_(extra["exp_avg_sq"])


def _flat_grad_base(eng, params, names, offs):
    """Device address of a flat gradient buffer in arena order if every .grad is a slice of one, else None."""
    g0 = params[0].grad
    if g0 is None or g0.dtype != torch.float32 or g0.device != eng.params.device:
        return None
    base = g0.data_ptr() - 4 * offs[names[0]]
    ok = all(p.grad is not None and p.grad.data_ptr() == base + 4 * offs[n] for p, n in zip(params, names))
    return base if ok else None


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(momentum=0.9, nesterov=True, weight_decay) with the two parameter groups of
    MMETrainingModule.py:17-20 (feature extractor at lr/3, classifier at lr), one HIP kernel per group over
    the flat arena (the groups are contiguous arena ranges: the classifier is laid out last)."""

    def __init__(self, owner: EngineOwner, lr_feature, lr_classifier, momentum=0.9, weight_decay=0.0):
        self.owner = owner
        params = owner._rln_params_in_arena_order()
        names = owner._rln_param_names
        fe = [p for p, n in zip(params, names) if n.startswith("featureExtractor.")]
        cl = [p for p, n in zip(params, names) if n.startswith("classifier.")]
        defaults = dict(lr=lr_classifier, momentum=momentum, weight_decay=weight_decay, nesterov=True)
        super().__init__([{"params": fe, "lr": lr_feature}, {"params": cl, "lr": lr_classifier}], defaults)
        eng = owner._rln_engine
        self.split = owner._rln_param_offsets["classifier.finalConv.weight"] if cl else eng.n_param
        self.buf = torch.zeros_like(eng.params)
        self._first = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self.owner._rln_sync()
        if self.buf.device != eng.params.device:
            self.buf = self.buf.to(eng.params.device)
        names = self.owner._rln_param_names
        offs = self.owner._rln_param_offsets
        params = [p for g in self.param_groups for p in g["params"]]
        base = _flat_grad_base(eng, params, names, offs)
        if base is None:
            for p, name in zip(params, names):
                gv = eng.grad_views[name]
                if p.grad is None:
                    gv.zero_()
                elif p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
            base = eng.grads.data_ptr()
        for g, (lo, hi) in zip(self.param_groups, [(0, self.split), (self.split, eng.n_param)]):
            if hi > lo:
                eng.sgd_step(self.buf, lo, hi, g["lr"], g["momentum"], g["weight_decay"], self._first, grads_ptr=base)
        self._first = False
        return loss
