"""Data-parallel training step of the hot path: one process per GPU, forward -> weighted CE -> backward
cut into segments whose finished gradient slices are all-reduced (RCCL over xGMI; backend "nccl" on ROCm)
on a side stream while the remaining backward runs -> fused AdamW on the flat arena.

Semantics follow un-synchronised Lightning DDP of the reference: gradients are averaged
over ranks; BatchNorm statistics, Dropout2d masks and class weights stay rank-local."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def plan_buckets(seg_ranges: Sequence[Tuple[int, int]], n_buckets: int = 4) -> List[Tuple[int, int, int, int]]:
    """Groups consecutive backward segments into ~equal-sized gradient buckets.

    seg_ranges[s] = (begin, end) of the arena slice segment s completes (segments run in index order and
    complete the arena back to front).  Returns [(seg_begin, seg_end, grad_begin, grad_end)]."""
    total = sum(e - b for b, e in seg_ranges)
    target = max(1, total // max(1, n_buckets))
    buckets, start, acc = [], 0, 0
    for s, (b, e) in enumerate(seg_ranges):
        acc += e - b
        last = s == len(seg_ranges) - 1
        if acc >= target or last:
            gb = min(r[0] for r in seg_ranges[start:s + 1])
            ge = max(r[1] for r in seg_ranges[start:s + 1])
            buckets.append((start, s + 1, gb, ge))
            start, acc = s + 1, 0
    return buckets


class BucketedGradReducer:
    """Runs ``run_segments(seg_begin, seg_end)`` bucket by bucket and overlaps the all-reduce of each finished
    gradient slice with the next bucket's backward.  Works on any flat tensor (CPU/gloo in tests)."""

    def __init__(self, flat_grads: torch.Tensor, seg_ranges, n_buckets=4, group=None, force_collectives=False):
        self.flat = flat_grads
        self.buckets = plan_buckets(seg_ranges, n_buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # force_collectives: run the bucketed all-reduce path even with one rank (plumbing check on a single GPU)
        self.force = bool(force_collectives) and dist.is_available() and dist.is_initialized()
        self.cuda = flat_grads.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grads.device) if self.cuda else None

    def backward_and_reduce(self, run_segments: Callable[[int, int], None], flat: Optional[torch.Tensor] = None):
        """flat: the gradient buffer of THIS backward when it is not the one given at construction (the module path
        hands every backward a fresh buffer)."""
        flat = self.flat if flat is None else flat
        if self.world == 1 and not self.force:
            run_segments(0, self.buckets[-1][1])
            return
        handles = []
        for (sb, se, gb, ge) in self.buckets:
            run_segments(sb, se)
            view = flat[gb:ge]
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            else:
                handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for h in handles:
            h.wait()
        # the sum is turned into the mean by grad_scale = 1/world in the optimiser kernel


class TrainStepper:
    """Fast path used by bench.py and scripts: bypasses autograd, drives the engine directly."""

    def __init__(self, engine, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, n_buckets=4,
                 force_collectives=False):
        self.eng = engine
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.m = torch.zeros_like(engine.params)
        self.v = torch.zeros_like(engine.params)
        self.step_no = 0
        self.reducer = BucketedGradReducer(engine.grads, engine.seg_ranges, n_buckets,
                                           force_collectives=force_collectives)
        self.world = self.reducer.world
        self.force = self.reducer.force

    def broadcast_parameters(self):
        """DDP's one-time parameter broadcast from rank 0."""
        if self.world > 1 or self.force:
            dist.broadcast(self.eng.params, 0)
            dist.broadcast(self.eng.bnrun, 0)

    def step(self, x, y, drop_scales=None, seed=None, lr=None):
        eng = self.eng
        probs, _ = eng.forward(x, training=True, with_backward=True, drop_scales=drop_scales, seed=seed)
        out, _, _ = eng.loss(probs, y, weighted=True)
        self.reducer.backward_and_reduce(lambda sb, se: eng.backward(1.0, sb, se))
        self.step_no += 1
        eng.adamw_step(self.m, self.v, self.step_no, self.lr if lr is None else lr, self.betas, self.eps, self.wd,
                       grad_scale=1.0 / self.world)
        return out  # device tensor: [loss, acc, bad_labels, counts...]; no host sync here
