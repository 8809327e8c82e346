"""Data-parallel gradient exchange for the training step: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference trains under nn.DataParallel: gradients are reduced onto GPU 0, Adam runs there and the 117 MB of
parameters are re-broadcast on every forward (tools/cp_train.py:115).  Here every rank keeps a replica and its own
Adam state; a step exchanges the flat gradient buffer of rsgnet_amd/optim.FusedAdam in a few large buckets.
xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE link, so buckets
are large (default 32 MB: 4 collectives for RSGNet-W32's 117 MB of fp32 gradients), launched on a side stream in
reverse parameter order so the first ones overlap the rest of the backward, and the 1/world average is folded
into the Adam kernel (`grad_scale`) instead of a pass over the buffer."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class FlatGradReducer:
    """All-reduces (sum) a flat gradient buffer in buckets.  `reduce()` returns after every bucket has been
    enqueued and the current stream waits on them; the caller applies 1/world (e.g. FusedAdam.grad_scale)."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 32 << 20, group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group)
        per = max(1, bucket_bytes // flat_grad.element_size())
        n = flat_grad.numel()
        # buckets from the END of the buffer: the last parameters get their gradients first in a backward pass
        self.buckets: List[torch.Tensor] = []
        self.ranges: List[tuple] = []
        hi = n
        while hi > 0:
            lo = max(0, hi - per)
            self.buckets.append(flat_grad[lo:hi])
            self.ranges.append((lo, hi))
            hi = lo
        self._stream = torch.cuda.Stream(flat_grad.device) if flat_grad.is_cuda else None
        self._done = [False] * len(self.buckets)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def begin(self):
        """Start of a step: no bucket has been exchanged yet."""
        self._done = [False] * len(self.buckets)

    def _launch(self, idx: List[int]):
        if not idx or self.world == 1:
            for k in idx:
                self._done[k] = True
            return
        if self._stream is None:                       # CPU tensors (gloo tests)
            works = [dist.all_reduce(self.buckets[k], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for k in idx]
            for w in works:
                w.wait()
        else:
            cur = torch.cuda.current_stream(self.flat.device)
            self._stream.wait_stream(cur)              # the gradients of these buckets are final on the current stream
            with torch.cuda.stream(self._stream):
                for k in idx:
                    dist.all_reduce(self.buckets[k], op=dist.ReduceOp.SUM, group=self.group)
        for k in idx:
            self._done[k] = True

    def ready(self, lo_elem: int):
        """Every gradient element at flat offset >= lo_elem is final (the backward pass has moved on to earlier
        parameters): the buckets that lie wholly in that tail start their all-reduce now, on the side stream, under the
        rest of the backward pass."""
        self._launch([k for k, (lo, _hi) in enumerate(self.ranges) if lo >= lo_elem and not self._done[k]])

    def reduce(self):
        """Exchange whatever `ready` has not started yet; the current stream then waits for every bucket."""
        self._launch([k for k in range(len(self.buckets)) if not self._done[k]])
        if self._stream is not None and self.world > 1:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)
        self._done = [False] * len(self.buckets)


class DataParallelStep:
    """One data-parallel training step per rank: the reference's loop body (lib/core/function.py:46-64: forward,
    criterion, zero_grad, backward, step) with the gradient exchange between backward and step.

    With `rsgnet_amd.optim.FusedAdam` the gradients already live in one flat buffer: it is all-reduced in place in
    buckets and the 1/world average rides in the Adam kernel (`grad_scale`).  Any other optimizer gets the generic
    form: gradients are flattened, summed over the ranks, averaged and copied back."""

    def __init__(self, model, criterion, optimizer, group: Optional[dist.ProcessGroup] = None, bucket_bytes: int = 32 << 20):
        self.model, self.criterion, self.optimizer, self.group = model, criterion, optimizer, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reducer: Optional[FlatGradReducer] = None
        flat = getattr(optimizer, "flat_grad", None)
        if flat is not None and self.world > 1:
            self.reducer = FlatGradReducer(flat, bucket_bytes, group)
            optimizer.grad_scale = self.reducer.grad_scale
            # overlap: the HIP backward pass (rsgnet_amd/train.py) reports after each of its segments which parameters
            # have their final gradient; the flat buffer keeps parameters in forward order, so what is final is a tail
            hooks = getattr(model, "grad_ready_hooks", None)
            if hooks is not None and hasattr(optimizer, "flat_offset_of"):
                hooks.clear()
                hooks.append(self._on_grads_ready)

    def _on_grads_ready(self, pending_params):
        """`pending_params`: the parameters whose gradient is NOT final yet."""
        cache = self.__dict__.setdefault("_ready_hi", {})          # the plan hands over the same list object every step
        key = (id(pending_params), len(pending_params), id(pending_params[0]) if pending_params else 0,
               id(pending_params[-1]) if pending_params else 0)
        hi = cache.get(key)
        if hi is None:
            hi = 0
            for p in pending_params:
                off = self.optimizer.flat_offset_of(p)
                if off is not None:
                    hi = max(hi, off + p.numel())
            cache[key] = hi
        self.reducer.ready(hi)

    def exchange(self):
        if self.world == 1:
            return
        if self.reducer is not None:
            self.reducer.reduce()
            return
        grads = [p.grad for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def __call__(self, x, *targets, **model_kwargs):
        """pose_hrnet: step(input, target, target_weight).  pose_rsgnet (rsgnet_train, function.py:256-323):
        step(input, target, target_weight, all_ins_target, all_ins_target_weight, target_limbs, relation_target=p) with
        criterion = lambda outs, *t: rsgnet_amd.loss.rsgnet_loss(outs, *t, JointsMSELoss(True))."""
        out = self.model(x, **model_kwargs)
        loss = self.criterion(out, *targets)
        self.optimizer.zero_grad()
        if self.reducer is not None:
            self.reducer.begin()
        loss.backward()
        self.exchange()
        self.optimizer.step()
        return loss, out
