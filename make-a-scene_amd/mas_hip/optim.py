"""``mas_hip.optim.Adam``: torch.optim.Adam's update (reference train.py:99-103) for fp32 CUDA parameters as ONE kernel launch per
step over all parameters of all groups that share hyper-parameters (``mas_adam_multi``).  Same arithmetic, same state layout
(``state[p] = {"step", "exp_avg", "exp_avg_sq"}``: ``state_dict()`` / ``load_state_dict()`` interchange with torch.optim.Adam), same
constructor arguments; ``amsgrad`` / ``maximize`` / ``capturable`` / ``differentiable`` are not implemented and raise.

It is a ``torch.optim.Optimizer``: the process-wide post-step hook of ``mas_hip.ops`` sees its steps, so packed weight images and
bf16 shadows are refreshed exactly as with torch's optimizers.  Parameters that are not fp32 CUDA tensors (none in the reference's
models) are updated by a plain torch expression with the same formula."""
import ctypes as C
import math

import torch

from . import AdamItem, check, lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False,
                 capturable=False, differentiable=False, fused=None, foreach=None):
        if amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("mas_hip.optim.Adam: amsgrad / maximize / capturable / differentiable are not implemented")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("mas_hip.optim.Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      capturable=False, differentiable=False, fused=None, foreach=None))
        self._tables = {}            # device index -> ring of (pinned staging, device table, event) slots + the bytes of the current table

    _RING = 4

    def _table(self, tkey, device, items):
        """The item table on the device (``tkey``: one ring of staging buffers per device, plus one per extra step count).  Gradient
        tensors are usually fresh allocations every step (``zero_grad(set_to_none=True)``), so the table changes every step: it goes through a ring of pinned staging buffers with an asynchronous copy each, and the
        host only waits for the copy issued ``_RING`` steps ago -- never for the step in flight (a wait on the previous step's copy
        would serialise the host behind the whole backward: measured, +9 ms of wall time per step)."""
        arr = (AdamItem * len(items))(*items)
        raw = bytes(memoryview(arr))
        ent = self._tables.get(tkey)
        if ent is not None and ent["raw"] == raw:
            return ent["slots"][ent["cur"]][1]
        nbytes = len(raw)
        if ent is None or ent["cap"] < nbytes:
            cap = max(nbytes, 1 << 16)
            ent = self._tables[tkey] = dict(cap=cap, cur=0, raw=None, slots=[
                (torch.empty(cap, dtype=torch.uint8, pin_memory=True), torch.empty(cap, dtype=torch.uint8, device=device), torch.cuda.Event())
                for _ in range(self._RING)], used=[False] * self._RING)
        ent["cur"] = (ent["cur"] + 1) % self._RING
        pinned, dev, ev = ent["slots"][ent["cur"]]
        if ent["used"][ent["cur"]]:
            ev.synchronize()         # the copy issued _RING table changes ago has left this staging buffer
        C.memmove(pinned.data_ptr(), raw, nbytes)
        dev[:nbytes].copy_(pinned[:nbytes], non_blocking=True)
        ev.record()
        ent["used"][ent["cur"]] = True
        ent["raw"] = raw
        return dev

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = float(group["lr"]), group["betas"], float(group["eps"]), float(group["weight_decay"])
            by_key, first = {}, {}          # (device, step number) -> items: parameters on different step counts take separate launches
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("mas_hip.optim.Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].device.type != "cpu":          # a state loaded from torch.optim.Adam(fused=True / capturable=True) keeps `step` on
                    st["step"] = st["step"].detach().to("cpu", torch.float32)   # the device: one sync here instead of one per step and parameter
                st["step"] += 1
                t = int(st["step"])
                if p.numel() == 0:
                    continue                                 # (torch.optim.Adam accepts empty parameters: nothing to update)
                native = (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                          and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous() and p.grad.device == p.device)
                if not native:
                    self._torch_update(p, st, t, lr, b1, b2, eps, wd)       # (another dtype / device)
                    continue
                key = (p.device, t)
                it = AdamItem()
                it.p, it.g, it.m, it.v, it.n = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                it.first_block = first.get(key, 0)
                first[key] = it.first_block + lib().mas_adam_blocks(p.numel())
                by_key.setdefault(key, []).append(it)
            for (device, t), items in by_key.items():
                with torch.cuda.device(device):
                    ordinal = [k for k in by_key if k[0] == device].index((device, t))      # 0 unless parameters of the group sit on different step counts
                    table = self._table((device.index, ordinal), device, items)
                    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                    check(lib().mas_adam_multi(C.c_void_p(table.data_ptr()), len(items), first[(device, t)], lr, b1, b2, eps, wd,
                                               1.0 - b1 ** t, 1.0 - b2 ** t, stream), "adam_multi")
        return loss

    @staticmethod
    def _torch_update(p, st, t, lr, b1, b2, eps, wd):
        g = p.grad if wd == 0.0 else p.grad.add(p, alpha=wd)
        st["exp_avg"].lerp_(g.to(st["exp_avg"].dtype), 1.0 - b1)
        st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (st["exp_avg_sq"].sqrt() / math.sqrt(1.0 - b2 ** t)).add_(eps)
        p.addcdiv_(st["exp_avg"], denom, value=-lr / (1.0 - b1 ** t))
