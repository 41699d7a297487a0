"""Data-parallel gradient averaging for the VQ / transformer training step: one process per GPU,
RCCL all-reduce over xGMI, a handful of large collectives per step.

The reference wraps the model in ``DistributedDataParallel`` (train.py:31-34); our modules keep working
under that wrapper (tests + ``bench.py --dp ddp``).  ``GradReducer`` is the opt-in lean equivalent used
by ``bench.py``: DDP copies-and-scales every parameter's gradient into its bucket with one tiny kernel
per parameter (345 launches per step for VQ-IMG, ~2 % of the step on an MI355X, plus the per-parameter
C++ hook); here a bucket is flattened by ONE batched-copy launch when its last gradient lands, averaged
by ONE asynchronous all-reduce (``ReduceOp.AVG`` on RCCL) that overlaps the rest of backward, and the
parameters' ``.grad`` are re-pointed at views of the reduced flat buffer -- no per-parameter kernels.

xGMI is point-to-point, so a ring all-reduce is per-link bound and wants few, large messages: the default
bucket is 128 MiB.  Gradients arrive in (roughly) reverse registration order, so every bucket but the one
holding the model's FIRST parameters closes while backward still has work to overlap with; that one closes
when backward ends and its collective is fully exposed -- it is therefore kept small (``first_bucket_bytes``,
8 MiB: 4 collectives for the 381 MB of VQ-IMG gradients, ~0.05 ms exposed instead of ~0.8 ms).

Semantics match DDP's: gradients are averaged over ranks; a parameter that received no gradient on this
rank contributes zeros (all ranks must agree on which parameters are trainable in a step, as with
``find_unused_parameters=False``); parameters are broadcast from rank 0 at construction.

Use it with ``optimizer.zero_grad(set_to_none=True)`` (the reference's and torch's default): autograd then hands each bucket fresh gradient
tensors and the bucket is flattened by one batched copy.  With ``set_to_none=False`` the gradients stay views of the flat buffer and
autograd ADDS the next step's gradients into them -- correct (the copy is skipped) but one in-place add launch per parameter: measured
+1.5 ms per VQ-IMG step (profiles/r05_dropin_defaults.txt).

Contract: ``finish()`` follows EVERY synchronising ``backward()``.  Gradient accumulation (the reference's
``accumulate_grad``, conf/img_config.yaml:13) runs the first k-1 micro-steps under ``with reducer.no_sync():``
(hooks idle, ``.grad`` accumulates locally as usual) and the k-th outside it, exactly like DDP's ``no_sync``.  A
second synchronising backward before ``finish()`` would add into a flat buffer whose all-reduce is in flight, so it
raises instead of producing racy, unreduced gradients."""
import contextlib
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

__all__ = ["GradReducer"]


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "wire", "pending", "work", "launched", "seen")

    def __init__(self, params: List[torch.nn.Parameter], wire_dtype: Optional[torch.dtype] = None):
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.flat = torch.zeros(off, dtype=params[0].dtype, device=params[0].device)
        # what crosses xGMI: the flat buffer itself, or (grad_dtype) a narrower copy of it -- the parameters' .grad stay fp32 views of `flat`
        self.wire = self.flat if wire_dtype in (None, params[0].dtype) else torch.zeros(off, dtype=wire_dtype, device=params[0].device)
        self.pending = len(params)
        self.work = None
        self.launched = False
        self.seen = set()


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = int(os.environ.get("MAS_DP_BUCKET_MB", "128")) << 20,
                 process_group: Optional[dist.ProcessGroup] = None, broadcast: bool = True,
                 first_bucket_bytes: int = 8 << 20, grad_dtype: Optional[torch.dtype] = None):
        """``grad_dtype`` (off by default; ``torch.bfloat16``): the gradients cross the links in that type -- one cast launch per bucket
        before the all-reduce, one back after it, half the bytes per step (190 MB instead of 381 MB for VQ-IMG); the sum over ranks is
        then formed in that type by the collective, so this is an A/B knob for the first multi-GPU run, not a default (``bench.py
        --grad-dtype bf16``).  ``bucket_bytes`` / ``first_bucket_bytes`` keep counting the fp32 gradients."""
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.avg_native = dist.get_backend(process_group) == "nccl"        # RCCL has ncclAvg; gloo does not
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("GradReducer: no trainable parameters")
        # contiguous runs of the registration order, split by size / dtype / device; the run holding the first parameters
        # (last to receive gradients) is capped at first_bucket_bytes.  Listed in reverse (~ the order they will close).
        groups, cur, cur_bytes = [], [], 0
        for p in plist:
            nb = p.numel() * p.element_size()
            cap = min(first_bucket_bytes, bucket_bytes) if not groups else bucket_bytes
            if cur and (cur_bytes + nb > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            groups.append(cur)
        self.grad_dtype = grad_dtype
        self.buckets: List[_Bucket] = [_Bucket(list(reversed(grp)), grad_dtype) for grp in reversed(groups)]
        self._where = {}
        self._hooks = []
        self._sync = True
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if broadcast:
            self.broadcast_parameters()

    # ------------------------------------------------------------------ #
    def broadcast_parameters(self, src: int = 0):
        """Rank ``src``'s parameter values to every rank (what DDP does at construction)."""
        with torch.no_grad():
            for b in self.buckets:
                flat = torch.cat([p.detach().reshape(-1) for p in b.params])
                dist.broadcast(flat, src, group=self.group)
                for p, off in zip(b.params, b.offsets):
                    p.copy_(flat[off:off + p.numel()].view_as(p))

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-steps of a gradient-accumulation window: backward passes inside the context only accumulate into
        ``.grad``; the first backward outside it reduces the accumulated sum (then call ``finish()``)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _on_grad(self, p: torch.nn.Parameter):
        if not self._sync:
            return
        b, i = self._where[p]
        if b.launched or i in b.seen:
            raise RuntimeError(
                "GradReducer: a second backward() reached a bucket whose all-reduce is already in flight. Call finish() "
                "after every synchronising backward(); run gradient-accumulation micro-steps under `with reducer.no_sync():`")
        b.seen.add(i)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        with torch.no_grad():
            pieces, in_place = [], 0
            base, esz = b.flat.data_ptr(), b.flat.element_size()
            for p, off in zip(b.params, b.offsets):
                g = p.grad
                if g is None:                                   # no gradient on this rank this step: zeros
                    g = torch.zeros_like(p, memory_format=torch.contiguous_format)
                elif g.data_ptr() == base + off * esz and g.is_contiguous():
                    in_place += 1                               # zero_grad(set_to_none=False): accumulated into last step's view
                pieces.append(g.contiguous().view(-1))
            if in_place != len(pieces):
                if in_place:                                    # mixed: the copy must not read what it overwrites
                    pieces = [g.clone() for g in pieces]
                torch.cat(pieces, out=b.flat)                   # one batched-copy launch per <=128 tensors
            if b.wire is not b.flat:
                b.wire.copy_(b.flat)                            # one cast launch; the collective below reads the narrow copy
            if self.avg_native:
                b.work = dist.all_reduce(b.wire, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                b.work = dist.all_reduce(b.wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for p, off in zip(b.params, b.offsets):             # .grad = view of the (soon) reduced flat buffer
                p.grad = b.flat[off:off + p.numel()].view_as(p)
            b.launched = True

    def finish(self):
        """Call after ``loss.backward()`` and before ``optimizer.step()``: closes buckets whose parameters did not
        all receive a gradient, waits for the collectives (the compute stream waits; the host does not block on RCCL)."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            b.work.wait()
            if b.wire is not b.flat:
                b.flat.copy_(b.wire)                            # back to the fp32 buffer the .grad views alias
            if not self.avg_native:
                b.flat.div_(self.world)
            b.work = None
            b.launched = False
            b.pending = len(b.params)
            b.seen.clear()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
