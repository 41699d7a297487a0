"""torch.autograd.Function wrappers over the C ABI (include/mas_hip.h).

Activations are torch tensors of LOGICAL shape [N,C,H,W] in ``channels_last`` memory
format (== the NHWC buffers the kernels expect), dtype bf16 or fp32.  Parameters stay
ordinary fp32 OIHW ``nn.Parameter``s (state_dict compatible with the reference); the
bf16 / NHWC-packed copies are a cache keyed on the parameter's stamp (in-place version, storage, optimizer steps: ``_param_stamp``).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import torch

from . import ACT_AFFINE, ACT_AFFINE_SILU, ACT_NONE, BF16, F32, WLAYOUT_K64, WLAYOUT_UP2, ConvDesc, PackItem, PackTileItem, check, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16}
_state = {"compute_dtype": torch.bfloat16 if os.environ.get("MAS_COMPUTE_DTYPE", "bf16") == "bf16" else torch.float32}


def compute_dtype() -> torch.dtype:
    return _state["compute_dtype"]


def set_compute_dtype(dt: torch.dtype) -> None:
    """bf16 (default; MFMA bf16, fp32 accumulate) or fp32 (exact-fp32 MFMA; parity mode)."""
    if dt not in _DT:
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    _state["compute_dtype"] = dt


_launch_hook = None


def set_launch_hook(fn) -> None:
    """fn(kind, shape, launch) wraps every conv launch (bench.py times the dominant kernel with HIP events
    recorded on the launch stream); shape = (n, h, w, cin, ho, wo, cout, ks, stride, act, has_residual); None disables it."""
    global _launch_hook
    _launch_hook = fn


def last_kernel() -> str:
    """the kernel the process's last launch ran (``mas_last_kernel``): the library dispatches on shape; tests assert the choice"""
    return lib().mas_last_kernel().decode()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the MI355X path needs a GPU tensor (no CPU fallback); got device {t.device}")


def nhwc(x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """[N,C,H,W]-logical tensor in channels_last memory, optionally cast."""
    if dtype is not None and x.dtype != dtype:
        return x.to(dtype=dtype, memory_format=torch.channels_last)
    return x.contiguous(memory_format=torch.channels_last)


def _empty_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


# --------------------------------------------------------------------------- #
# packed-weight cache
# --------------------------------------------------------------------------- #
# Validity of every derived copy of a parameter (packed conv weights, bf16 Linear shadows): ``(Parameter._version, data_ptr,
# optimizer generation)``.  ``_version`` alone is NOT enough: the fused optimizers (``torch.optim.Adam(fused=True)`` and friends)
# update parameters through a multi-tensor kernel that does not bump it (measured on this torch build: version 0 -> 0 with the
# data changed), so a process-wide optimizer post-step hook counts, per parameter, the optimizer steps that touched it.
_param_generation = {}


_param_generation_refs = {}


def _note_optimizer_step(optimizer, *_args, **_kwargs):
    for group in optimizer.param_groups:
        for p in group["params"]:
            k = id(p)
            if k not in _param_generation_refs:           # the counter dies with the parameter (ids are reused by CPython)
                _param_generation_refs[k] = weakref.ref(p, lambda _r, k=k: (_param_generation.pop(k, None), _param_generation_refs.pop(k, None)))
            _param_generation[k] = _param_generation.get(k, 0) + 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_hook
    _register_post_hook(_note_optimizer_step)
except ImportError:                                   # older torch: fall back to wrapping Optimizer.step once
    _orig_opt_step = torch.optim.Optimizer.step

    def _step_and_note(self, *a, **k):
        out = _orig_opt_step(self, *a, **k)
        _note_optimizer_step(self)
        return out

    torch.optim.Optimizer.step = _step_and_note


def _param_stamp(p: torch.Tensor):
    return (p._version, p.data_ptr(), _param_generation.get(id(p), 0))


# debugging aid for writes the stamp cannot see (``w.data.copy_(...)`` without invalidate_weight_cache()): every cache hit compares a
# checksum of the live parameter with the one taken when its image was packed (one device sync per conv call: never on by default)
_CACHE_CHECK = os.environ.get("MAS_WEIGHT_CACHE_CHECK", "0") == "1"


def _checksum(w: torch.Tensor) -> float:
    return float(w.detach().double().sum())


class _PackCache:
    """bf16/NHWC-packed copies of conv parameters, valid while the parameter's stamp (``_param_stamp``: in-place version, storage,
    optimizer steps) is unchanged.  Only ``nn.Parameter`` objects are cached (held by weak reference); temporaries (e.g. the
    concatenated q|k|v weight) are packed on every call."""

    def __init__(self):
        self.store = {}
        self.derived = {}                            # images of tensors DERIVED from several parameters (AttnBlock's q|k|v stack)
        self.sums = {}                               # MAS_WEIGHT_CACHE_CHECK=1 only
        self._tables = {}                            # device-resident item tables of the batched pack launches

    def clear(self):
        self.store.clear()
        self.derived.clear()
        self._tables.clear()

    def drop(self, w: torch.Tensor):
        """forget every image made from parameter ``w`` (its module was switched train()/eval() or reloaded)"""
        k = id(w)
        for key in [key for key in self.store if key[0] == k]:
            self.store.pop(key, None)
        for key in [key for key in self.derived if k in key[0]]:
            self.derived.pop(key, None)

    def get_derived(self, w: torch.Tensor, sources, transpose: bool, dtype: torch.dtype, layout: int) -> torch.Tensor:
        """``w`` is a function of the parameters ``sources`` only (e.g. their concatenation): its packed image is valid while all
        of their stamps are unchanged."""
        key = (tuple(id(p) for p in sources), transpose, dtype, layout)
        stamp = tuple(_param_stamp(p) for p in sources)
        hit = self.derived.get(key)
        if hit is not None and hit[1] == stamp and all(r() is p for r, p in zip(hit[0], sources)) and hit[2].device == w.device:
            return hit[2]
        packed = pack_conv_weight(w.detach(), transpose, dtype, layout)
        refs = tuple(weakref.ref(p, lambda _r, k=key: self.derived.pop(k, None)) for p in sources)
        self.derived[key] = (refs, stamp, packed)
        return packed

    def get(self, w: torch.Tensor, transpose: bool, dtype: torch.dtype, layout: int = WLAYOUT_K64, sources=None) -> torch.Tensor:
        if not isinstance(w, torch.nn.Parameter):
            if sources is not None:
                return self.get_derived(w, sources, transpose, dtype, layout)
            return pack_conv_weight(w.detach(), transpose, dtype, layout)
        key = (id(w), transpose, dtype, layout)
        hit = self.store.get(key)
        if hit is not None and hit[0]() is w and hit[1] == _param_stamp(w):
            if _CACHE_CHECK and self.sums.get(key) != _checksum(w):
                raise RuntimeError("packed-weight cache: parameter of shape %s changed without a version bump / optimizer step (a write "
                                   "through .data?) -- call mas_hip.ops.invalidate_weight_cache() after such writes" % (tuple(w.shape),))
            return hit[2]
        if hit is None or hit[0]() is not w or hit[2].device != w.device:
            n = _packed_elems(w.shape[0], w.shape[1], w.shape[2], layout)
            self.store[key] = [weakref.ref(w, lambda _r, k=key: self.store.pop(k, None)), None, torch.empty(n, dtype=dtype, device=w.device)]
        self._refresh_stale(w.device)
        return self.store[key][2]

    def _refresh_stale(self, device):
        """Repacks EVERY stale entry on ``device``: after an optimizer step all of a model's images are stale at once.  The bf16 images
        of a parameter (forward / data-gradient operand, K64 / K32) are written together from ONE tiled read of it
        (``mas_pack_conv_weight_tiles``: one launch for the whole model); fp32 images (the parity mode) keep the gather kernel
        (``mas_pack_conv_weight_batch``, also one launch).  The packed buffers are refreshed in place (nothing saves them for backward:
        the backward asks the cache again)."""
        groups, items, first, fresh, keep = {}, [], 0, [], []
        for key, ent in list(self.store.items()):                                  # (a weakref callback may pop entries meanwhile)
            wid, transpose, dtype, layout = key
            w = ent[0]()
            if w is None or w.device != device or ent[1] == _param_stamp(w):
                continue
            cout, cin, ks, _ = w.shape
            wf = w.detach()
            if wf.dtype != torch.float32 or not wf.is_contiguous():
                wf = wf.contiguous().float()
                keep.append(wf)                                # keep the temporary alive until the launches have been issued
            ent[1] = None                                      # not valid until the launch below has been accepted
            fresh.append((ent, _param_stamp(w)))
            if _CACHE_CHECK:
                self.sums[key] = _checksum(w)
            g = groups.get(wid)
            if dtype == torch.bfloat16 and ks <= 4 and layout != WLAYOUT_UP2 and (g is None or len(g[2]) < 4):
                if g is None:
                    g = groups[wid] = (wf, (cout, cin, ks), [])
                g[2].append((ent[2].data_ptr(), int(transpose), int(layout)))
                continue
            nb = lib().mas_pack_batch_blocks(cout, cin, ks, int(transpose), _DT[dtype], int(layout))
            if nb <= 0:
                raise RuntimeError(f"pack_conv_weight: unsupported weight shape {tuple(w.shape)}")
            items.append(PackItem(wf.data_ptr(), ent[2].data_ptr(), cout, cin, ks, int(transpose), _DT[dtype], int(layout), first, nb))
            first += nb
        if groups:
            titems, tfirst, max_ks = [], 0, 1
            for wf, (cout, cin, ks), imgs in groups.values():
                it = PackTileItem()
                it.w_oihw, it.n_img, it.Cout, it.Cin, it.ks, it.first_block = wf.data_ptr(), len(imgs), cout, cin, ks, tfirst
                for k, (ptr, tr, lay) in enumerate(imgs):
                    it.img[k], it.transpose[k], it.layout[k] = ptr, tr, lay
                titems.append(it)
                tfirst += lib().mas_pack_tile_blocks(cout, cin, ks)
                max_ks = max(max_ks, ks)
            table = self._upload("tiles", device, (PackTileItem * len(titems))(*titems))
            check(lib().mas_pack_conv_weight_tiles(_ptr(table), len(titems), tfirst, max_ks, _stream()), "pack_conv_weight_tiles")
        if items:
            table = self._upload("batch", device, (PackItem * len(items))(*items))
            check(lib().mas_pack_conv_weight_batch(_ptr(table), len(items), first, _stream()), "pack_conv_weight_batch")
        for ent, stamp in fresh:                                 # (a failed launch raised above: the entries stay invalid)
            ent[1] = stamp

    def _upload(self, which, device, arr):
        """the item table on the device; pointers and shapes repeat step after step: one upload, then reuse"""
        raw = bytes(memoryview(arr))
        sig = (device.index, raw)
        hit = self._tables.get(which)
        if hit is None or hit[0] != sig:
            hit = self._tables[which] = (sig, torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device))
        return hit[1]


_pack_cache = _PackCache()


def drop_weight_cache_of(*params) -> None:
    """Drops the packed images / bf16 shadows made from these parameters only (``Conv2d`` / ``Linear`` call it for their own
    parameters on load_state_dict and on a train()/eval() switch -- not the whole process's cache per child module)."""
    for p in params:
        if p is not None:
            _pack_cache.drop(p)
            _bf16_shadows.drop(p)


def invalidate_weight_cache() -> None:
    """Drops every cached packed weight / bf16 shadow.  The caches are validated by ``_param_stamp`` (``Parameter._version``,
    ``data_ptr``, optimizer steps seen by the global post-step hook), which an in-place write THROUGH ``.data`` (``w.data.copy_(ema)``, ``w.data.normal_()``, weight clipping) does not change: call this after
    such a write (``MAS_WEIGHT_CACHE_CHECK=1`` makes every conv-weight hit compare a checksum of the parameter against the one taken at
    pack time and raise on a mismatch: a debugging aid for exactly this).  ``models.modules.Conv2d`` drops ITS OWN entries
    (``drop_weight_cache_of``) from ``load_state_dict`` and on every train()/eval() switch."""
    _pack_cache.clear()
    _bf16_shadows.clear()


def _packed_elems(cout: int, cin: int, ks: int, layout: int) -> int:
    """elements of a packed image: the sub-pixel image of Upsample + conv (UP2) holds 4 phases x 4 taps instead of 9 taps"""
    if layout == WLAYOUT_UP2:
        return int(lib().mas_packed_weight_elems_up2(cout, cin))
    return int(lib().mas_packed_weight_elems(cout, cin, ks))


def pack_conv_weight(w: torch.Tensor, transpose: bool, dtype: torch.dtype, layout: int = WLAYOUT_K64) -> torch.Tensor:
    """OIHW fp32 -> the LDS image of ``layout`` (include/mas_hip.h: K64 = every kernel but the wide 3x3 one, K32 = that one)."""
    _require_cuda(w, "pack_conv_weight")
    cout, cin, ks, ks2 = w.shape
    assert ks == ks2
    w = w.contiguous().float()
    n = _packed_elems(cout, cin, ks, layout)
    out = torch.empty(n, dtype=dtype, device=w.device)
    check(lib().mas_pack_conv_weight_layout(_ptr(w), _ptr(out), cout, cin, ks, int(transpose), _DT[dtype], int(layout), _stream()),
          "pack_conv_weight")
    return out


class ConvWeight:
    """An UNPACKED conv weight handed to ``conv_fwd_raw``: packed there (through the cache when it is an nn.Parameter) in the
    layout the library prefers for that convolution (``mas_conv_weight_layout``)."""
    __slots__ = ("w", "transpose", "sources")

    def __init__(self, w: torch.Tensor, transpose: bool = False, sources=None):
        # sources: the nn.Parameters a non-Parameter ``w`` was built from (their concatenation, say): its packed image is then cached
        # on THEIR stamps instead of being re-packed on every call
        self.w, self.transpose, self.sources = w, bool(transpose), sources


# --------------------------------------------------------------------------- #
# raw kernel calls
# --------------------------------------------------------------------------- #
# Fused GroupNorm statistics (mas_conv_fwd_stats): ON by default since round 3 (MAS_FUSED_GN_STATS=0 turns them off).  The wide
# kernel's epilogue adds, per lane, the sums / sums of squares of its 4 output channels on the packed-fp32 instructions (values
# before the bf16 rounding) and writes one table row per tile; the consumer's GroupNorm then needs only the finalize launch instead
# of a pass over the tensor.  Round 2 measured this neutral (the producers were the GroupNorm+SiLU-loader variants of the kernel,
# 17+ spilled registers: +0.07 ms per launch against the 0.10 ms pass it removes); with the activation materialised in training
# (MAS_GN_MATERIALIZE) the producers are the prologue-free variants (no spills): +0.016 ms per launch, 64.47 -> 63.73 ms per step
# (profiles/r03_ab_v4.txt).
_stats_state = {"on": os.environ.get("MAS_FUSED_GN_STATS", "1") == "1", "stash": None}


def _take_stats(x: torch.Tensor):
    """(partial table, rows per image) the convolution that produced ``x`` left for its consumer's GroupNorm, or (None, 0).
    The table rides on the tensor OBJECT (``x._mas_gn``, set by ``norm_act_conv`` / ``resblock``): a copy, a cast or an in-place
    write (``_version``) silently drops it and the statistics are recomputed from the tensor."""
    st = getattr(x, "_mas_gn", None)
    if st is None or st[2] != x._version or not x.is_contiguous(memory_format=torch.channels_last):
        return None, 0
    return st[0], st[1]


def _attach_stats(y: torch.Tensor):
    st, _stats_state["stash"] = _stats_state["stash"], None
    if st is not None:
        y._mas_gn = (st[0], st[1], y._version)
    return y


def gn_stats(x: torch.Tensor, gamma, beta, groups: int, eps: float, partial: Optional[torch.Tensor] = None, rows: int = 0):
    """x channels_last [N,C,H,W] -> (mean_rstd [N,G,2], scale_shift [N,C,2]) fp32.  With ``partial`` (the per-tile sums the producing
    convolution wrote, ``mas_conv_fwd_stats``) the pass over the tensor is skipped: only the finalize runs."""
    n, c, h, w = x.shape
    mr = torch.empty((n, groups, 2), dtype=torch.float32, device=x.device)
    ss = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
    if partial is not None:
        check(lib().mas_gn_stats_from_partials(_ptr(partial), n, h * w, c, groups, int(rows), float(eps), _ptr(gamma), _ptr(beta), _ptr(mr),
                                               _ptr(ss), _stream()), "gn_stats_from_partials")
        return mr, ss
    wsb = lib().mas_gn_stats_workspace(n, c)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
    check(lib().mas_gn_stats(_ptr(x), _DT[x.dtype], n, h * w, c, groups, float(eps), _ptr(gamma), _ptr(beta), _ptr(mr), _ptr(ss),
                             _ptr(ws), wsb, _stream()), "gn_stats")
    return mr, ss


# Materialised activation (mas_gn_act): in TRAINING the GroupNorm(+SiLU) output of a 3x3 layer is written once (one read + one write of
# the tensor, 0.23 ms at 128 ch @256^2 x 32) and both its consumers -- the forward convolution and, in the backward, the weight
# gradient -- run prologue-free on it, instead of each recomputing the activation in its loader (+0.18 ms and +0.145 ms at that
# shape).  Under torch.no_grad() (inference, the encoder-stack line of bench.py) there is no second consumer and the fused loader
# stays.  MAS_GN_MATERIALIZE=0: fused loaders everywhere (round 2's scheme).  A/B: profiles/r03_ab_v3.txt.
_MATERIALIZE = os.environ.get("MAS_GN_MATERIALIZE", "1") == "1"
# MAS_SAVE_ACT=0 (``set_save_activations(False)``): the materialised activation is still written and read by the forward convolution
# but NOT kept for the backward -- the weight gradient then recomputes it in its loader from x and the scale / shift table (the fused
# prologue of mas_conv_wgrad: +0.145 ms per 128-channel 256^2 launch), and the allocator's high-water mark drops by the activations'
# 12 GiB at batch 32 (one bf16 tensor per GroupNorm-fed convolution).  Default: keep them (288 GB of HBM; the step is 1 ms faster).
_save_act = {"on": os.environ.get("MAS_SAVE_ACT", "1") == "1"}


def set_save_activations(on: bool) -> None:
    _save_act["on"] = bool(on)


def _gn_act_ok(c: int, dtype: torch.dtype) -> bool:
    epu = 8 if dtype == torch.bfloat16 else 4
    return c % epu == 0 and 256 % (c // epu) == 0 and getattr(lib(), "mas_gn_act", None) is not None


def gn_act(x: torch.Tensor, ss: torch.Tensor, act: int) -> torch.Tensor:
    """act(x * scale + shift) as a tensor (channels_last, x's dtype): ``mas_gn_act``."""
    n, c, h, w = x.shape
    a = torch.empty_like(x, memory_format=torch.channels_last)
    check(lib().mas_gn_act(_ptr(x), _ptr(a), _DT[x.dtype], n, h * w, c, act, _ptr(ss), _stream()), "gn_act")
    return a


_small_memo = {}


def gn_small_ok(x: torch.Tensor, groups: int) -> bool:
    """does the tensor take the small-map GroupNorm kernels (``mas_gn_small_supported``: bf16, h*w <= 1024, C % 64 == 0)?"""
    n, c, h, w = x.shape
    key = (x.dtype, h * w, c, groups)
    ok = _small_memo.get(key)
    if ok is None:
        ok = _small_memo[key] = bool(x.dtype in _DT and lib().mas_gn_small_supported(_DT[x.dtype], h * w, c, groups))
    return ok


def gn_stats_act(x: torch.Tensor, gamma, beta, groups: int, eps: float, act: int):
    """(mean_rstd, scale_shift, act(gn(x))) in ONE launch for small maps (``mas_gn_stats_act``; see ``gn_small_ok``): replaces
    ``gn_stats`` + ``gn_act`` (three dependent launches) where the producing convolution left no fused statistics."""
    n, c, h, w = x.shape
    mr = torch.empty((n, groups, 2), dtype=torch.float32, device=x.device)
    ss = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
    a = torch.empty_like(x, memory_format=torch.channels_last)
    check(lib().mas_gn_stats_act(_ptr(x), _ptr(a), _DT[x.dtype], n, h * w, c, groups, float(eps), _ptr(gamma), _ptr(beta), act, _ptr(mr),
                                 _ptr(ss), _stream()), "gn_stats_act")
    return mr, ss, a


def _gn_bwd_aten(x, da, dres, groups, act, gamma, mr, ss):
    """The same backward from ATen element-wise ops and reductions ON THE GPU, for channel counts the streaming kernels do not take (their
    threads keep one 16-byte channel slot across the grid stride: C / 8 must divide 256 -- every width of the reference's configs does;
    192, 320, ... do not).  Off the benched path; found by tests/test_gpu_conv_random.py (round 6).  u is rebuilt from the saved
    scale / shift table, xhat from the saved (mean, rstd)."""
    n, c, h, w = x.shape
    cg = c // groups
    xf, df = x.float(), da.float()
    u = xf * ss[..., 0].view(n, c, 1, 1) + ss[..., 1].view(n, c, 1, 1)
    if act == ACT_AFFINE_SILU:
        sg = torch.sigmoid(u)
        df = df * (sg * (1.0 + u * (1.0 - sg)))
    mean, rstd = mr[..., 0].view(n, groups, 1, 1, 1), mr[..., 1].view(n, groups, 1, 1, 1)
    xhat = (xf.reshape(n, groups, cg, h, w) - mean) * rstd
    g = (df * gamma.float().view(1, c, 1, 1)).reshape(n, groups, cg, h, w)
    m = float(cg * h * w)
    s1, s2 = g.sum((2, 3, 4), keepdim=True) / m, (g * xhat).sum((2, 3, 4), keepdim=True) / m
    dx = (rstd * (g - s1 - xhat * s2)).reshape(n, c, h, w)
    if dres is not None:
        dx = dx + dres.float()
    dgamma = (df.reshape(n, groups, cg, h, w) * xhat).reshape(n, c, h, w).sum((0, 2, 3))
    dbeta = df.sum((0, 2, 3))
    return dx.to(x.dtype).contiguous(memory_format=torch.channels_last), dgamma, dbeta


def gn_bwd(x, da, dres, groups, act, gamma, mr, ss, path=None):
    """GroupNorm(+SiLU) backward: (dx [+ dres], dgamma, dbeta).  ``path`` None: ``mas_gn_bwd`` (the small-map kernel up to 512 pixels,
    bf16; else the three launches); "three": reduce / finalize / apply launches on any shape (``mas_gn_bwd_3pass``).  Channel counts
    outside the kernels' envelope (``_gn_act_ok``) take ``_gn_bwd_aten``."""
    n, c, h, w = x.shape
    if not _gn_act_ok(c, x.dtype):
        return _gn_bwd_aten(x, da, dres, groups, act, gamma, mr, ss)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    wsb = lib().mas_gn_bwd_workspace(n, c)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
    fn = {None: lib().mas_gn_bwd, "three": lib().mas_gn_bwd_3pass}[path]
    check(fn(_ptr(x), _ptr(da), _ptr(dres), _DT[x.dtype], n, h * w, c, groups, act, _ptr(gamma), _ptr(mr), _ptr(ss),
             _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), wsb, _stream()), "gn_bwd")
    return dx, dgamma, dbeta


def _desc(n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, in_dt, out_dt, act, upsample, w_layout=WLAYOUT_K64):
    return ConvDesc(n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, _DT[in_dt], _DT[out_dt], act, int(upsample), int(w_layout))


_layout_memo = {}


def _preferred_layout(d: ConvDesc) -> int:
    key = tuple(getattr(d, f) for f, _ in ConvDesc._fields_[:-1])
    lay = _layout_memo.get(key)
    if lay is None:
        lay = _layout_memo[key] = int(lib().mas_conv_weight_layout(C.byref(d)))
    return lay


_stat_rows_memo = {}
_up2_memo = {}


def _dev_key():
    return torch.cuda.current_device()               # eligibility depends on the device's CU count (mas_num_cus)


def _up2_wgrad_splits(d: ConvDesc) -> int:
    key = ("w", _dev_key()) + tuple(getattr(d, f) for f, _ in ConvDesc._fields_[:-1])
    k = _up2_memo.get(key)
    if k is None:
        k = _up2_memo[key] = int(lib().mas_conv_up2_wgrad_splits(C.byref(d)))
    return k


def _up2_supported(d: ConvDesc, dgrad: bool = False) -> bool:
    key = (dgrad, _dev_key()) + tuple(getattr(d, f) for f, _ in ConvDesc._fields_[:-1])
    ok = _up2_memo.get(key)
    if ok is None:
        fn = lib().mas_conv_up2_dgrad_supported if dgrad else lib().mas_conv_up2_supported
        ok = _up2_memo[key] = bool(fn(C.byref(d)))
    return ok


# The fast kernels (wide / stream / stride-2 / thin / 1x1 convolutions, LDS-DMA weight gradients) address a whole tensor through ONE
# buffer descriptor with 31-bit byte offsets, and refuse tensors of 2^31 bytes or more -- 128 ch @256^2 bf16 reaches that at N = 128,
# and the library then quietly takes round 1's kernels (20-30 % slower; VERDICT r3 "missing" 3: the reference hints at 192 images per
# GPU, conf/img_config.yaml:17).  Images are independent, so such a launch is cut into batch slices below the limit here: every slice
# runs the kernel the shape deserves, outputs / statistics rows / split-K slabs land at their offsets in the full-size buffers.
_MAX_TENSOR_BYTES = (1 << 31) - 1


def _batch_slices(n, *bytes_per_image):
    """[(n0, n1), ...] such that every listed tensor of a slice stays under 2^31 bytes (one slice when the whole batch does)"""
    worst = max(bytes_per_image)
    if n * worst <= _MAX_TENSOR_BYTES or n <= 1:
        return [(0, n)]
    per = max(1, _MAX_TENSOR_BYTES // worst)
    k = -(-n // per)
    per = -(-n // k)                                  # even slices
    return [(a, min(n, a + per)) for a in range(0, n, per)]


def conv_fwd_raw(x, ss, wp, bias, residual, n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, act, upsample, out_dtype, want_stats=False):
    """``wp``: a ``ConvWeight`` (packed here in the layout the library prefers for this convolution) or an already packed
    K64 image from ``pack_conv_weight`` (always accepted; the call then stays on the kernels that read K64).
    ``want_stats``: returns (y, partial, rows) -- the per-tile channel sums of y for the GroupNorm that consumes it
    (``mas_conv_fwd_stats``), or (y, None, 0) when this launch has no fused statistics: kernels other than the wide one, and the
    wide kernel's GroupNorm+SiLU-loader variants (they are at the register limit: the statistics epilogue costs them +0.07 ms per launch
    against the 0.10 ms pass it removes, with 17 more spilled registers -- profiles/r03_kernel_trace_encoder_fwd.txt)."""
    y = _empty_nhwc(n, cout, ho, wo, out_dtype, x.device)
    esz, osz = x.element_size(), y.element_size()
    slices = _batch_slices(n, h * w * cin * esz, ho * wo * cout * osz)
    ns = slices[0][1] - slices[0][0]
    d = _desc(ns, h, w, cin, ho, wo, cout, ks, stride, pt, pl, x.dtype, out_dtype, act, upsample)
    if isinstance(wp, ConvWeight):
        d.w_layout = _preferred_layout(d)
        # Upsample + conv in its sub-pixel form (conv_up2.hip): 2.25x fewer FLOPs.  Eligibility depends on N (tile count against the
        # CU count): EVERY batch slice has to take the kernel, or the whole launch keeps the 3x3 image (a smaller last slice must not
        # turn into "w_layout UP2 but ... does not take the sub-pixel kernel")
        if upsample and residual is None and act == ACT_NONE and all(
                _up2_supported(d if n1 - n0 == ns else _desc(n1 - n0, h, w, cin, ho, wo, cout, ks, stride, pt, pl, x.dtype, out_dtype, act, upsample))
                for n0, n1 in slices):
            d.w_layout = WLAYOUT_UP2
        wp = _pack_cache.get(wp.w, wp.transpose, x.dtype, d.w_layout, wp.sources)
    partial, rows = None, 0
    if want_stats and _stats_state["on"] and act == ACT_NONE:
        key = tuple(getattr(d, f) for f, _ in ConvDesc._fields_)
        rows = _stat_rows_memo.get(key)
        if rows is None:
            rows = _stat_rows_memo[key] = int(lib().mas_conv_stat_rows(C.byref(d)))
        if rows > 0:
            partial = torch.empty(n * rows * cout * 2, dtype=torch.float32, device=x.device)

    def launch():
        for n0, n1 in slices:
            ds = d if n1 - n0 == ns else _desc(n1 - n0, h, w, cin, ho, wo, cout, ks, stride, pt, pl, x.dtype, out_dtype, act, upsample, d.w_layout)
            xs, ys = x[n0:n1], y[n0:n1]
            sss = ss[n0:n1] if ss is not None else None
            rs = residual[n0:n1] if residual is not None else None
            if partial is not None:
                ps = partial[n0 * rows * cout * 2:]
                check(lib().mas_conv_fwd_stats(C.byref(ds), _ptr(xs), _ptr(sss), _ptr(wp), _ptr(bias), _ptr(rs), _ptr(ys), _ptr(ps), _stream()),
                      "conv_fwd_stats")
            else:
                check(lib().mas_conv_fwd(C.byref(ds), _ptr(xs), _ptr(sss), _ptr(wp), _ptr(bias), _ptr(rs), _ptr(ys), _stream()), "conv_fwd")

    if _launch_hook is not None:
        _launch_hook("conv_fwd", (n, h, w, cin, ho, wo, cout, ks, stride, act, int(residual is not None)), launch)
    else:
        launch()
    if want_stats:
        return y, partial, (rows if partial is not None else 0)
    return y


def conv_wgrad_raw(x, ss, dy, n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, act, upsample, want_bias):
    if ks == 4 and stride == 2:
        # the discriminator's 4x4 / stride-2 convolutions (reference losses/discriminator.py:20,27): the stride-2 4x4 correlation of x
        # is the stride-1 2x2 correlation of its (padded) space-to-depth image, so the weight gradient runs on the stride-1
        # transpose-read kernel with ks = 2 and 4 Cin channels; dW' [co][(dy,dx,c)][kh'][kw'] -> dW [co][c][2kh'+dy][2kw'+dx]
        if act != ACT_NONE or upsample or pt != pl:
            raise RuntimeError("conv_wgrad: the 4x4 stride-2 geometry takes no prologue / upsample fold and needs pad_top == pad_left")
        xs = space_to_depth2x(x, ho + 1, wo + 1, pt)
        dws, db = conv_wgrad_raw(xs, None, dy, n, ho + 1, wo + 1, 4 * cin, ho, wo, cout, 2, 1, 0, 0, ACT_NONE, False, want_bias)
        dw = dws.view(cout, 2, 2, cin, 2, 2).permute(0, 3, 4, 1, 5, 2).reshape(cout, cin, 4, 4)
        return dw, db
    nw = cout * ks * ks * cin
    esz = x.element_size()
    slices = _batch_slices(n, h * w * cin * esz, ho * wo * cout * esz)        # (see conv_fwd_raw: tensors of 2^31 bytes and more)
    if upsample and ks == 3 and stride == 1 and act == ACT_NONE and ss is None and x.dtype == torch.bfloat16:
        # Upsample + conv: the weight gradient in the sub-pixel form too (conv_wgrad_dma.hip, 2x2 taps per phase): 2.25x fewer MFMAs for the
        # same bytes; mas_wgrad_reduce_up2 folds the 4 x 4 phase taps back into the 3x3 gradient (fixed order: bitwise reproducible).
        # Batch slices (tensors of 2^31 bytes and more) are reduced one by one and added in slice order.
        ds = [_desc(n1 - n0, h, w, cin, ho, wo, cout, ks, 1, pt, pl, x.dtype, x.dtype, ACT_NONE, True) for n0, n1 in slices]
        ks_ = [_up2_wgrad_splits(d) for d in ds]
        if all(k > 0 for k in ks_):
            nw4 = cout * 4 * cin
            need = max(4 * k * (nw4 + cout) for k in ks_)
            key = (x.device.index, torch.cuda.current_stream().cuda_stream)
            ws = _wgrad_partials.get(key)
            if ws is None or ws.numel() < need:
                ws = _wgrad_partials[key] = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=x.device)
            dwo = db = None
            for (n0, n1), d, k in zip(slices, ds, ks_):
                pb0 = ws.data_ptr() + 4 * 4 * k * nw4
                dws = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
                dbs = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
                xs, dys = (x, dy) if len(slices) == 1 else (x[n0:n1], dy[n0:n1])

                def launch_up2():
                    check(lib().mas_conv_up2_wgrad_partial(C.byref(d), _ptr(xs), _ptr(dys), _ptr(ws), C.c_void_p(pb0) if want_bias else None, _stream()),
                          "conv_up2_wgrad_partial")

                if _launch_hook is not None:
                    _launch_hook("conv_wgrad", (n1 - n0, h, w, cin, ho, wo, cout, ks, stride, act, 0), launch_up2)
                else:
                    launch_up2()
                check(lib().mas_wgrad_reduce_up2(_ptr(ws), C.c_void_p(pb0) if want_bias else None, k, _ptr(dws), _ptr(dbs), cout, cin, _stream()),
                      "wgrad_reduce_up2")
                dwo = dws if dwo is None else dwo.add_(dws)
                if want_bias:
                    db = dbs if db is None else db.add_(dbs)
            return dwo, db
    descs = [_desc(n1 - n0, h, w, cin, ho, wo, cout, ks, stride, pt, pl, x.dtype, x.dtype, act, upsample) for n0, n1 in slices]
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    splits = [int(lib().mas_conv_wgrad_splits(C.byref(d))) for d in descs]
    shape = (n, h, w, cin, ho, wo, cout, ks, stride, act, 0)
    dwo = torch.empty((cout, cin, ks, ks), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    if all(k > 0 for k in splits):
        # The convolutions that carry the FLOPs: every split-K work-group stores its partial sums into its own slab of a persistent
        # workspace (written in full by each launch: never zeroed) and ``mas_wgrad_reduce`` adds the slabs in a fixed order straight
        # into the OIHW gradient.  No fp32 atomics (they cost 45-60 us per launch and made the sums order-dependent): the weight
        # gradient of these layers is bitwise reproducible run to run.  Batch slices append their slabs: one reduce over all of them.
        tot = sum(splits)
        need = tot * (nw + cout)
        ws = _wgrad_partials.get(key)
        if ws is None or ws.numel() < need:
            ws = _wgrad_partials[key] = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=x.device)
        pb0 = ws.data_ptr() + 4 * tot * nw

        def launch():
            first = 0
            for (n0, n1), d, k in zip(slices, descs, splits):
                check(lib().mas_conv_wgrad_partial(C.byref(d), _ptr(x[n0:n1]), _ptr(ss[n0:n1] if ss is not None else None), _ptr(dy[n0:n1]),
                                                   C.c_void_p(ws.data_ptr() + 4 * first * nw), C.c_void_p(pb0 + 4 * first * cout) if want_bias else None,
                                                   _stream()), "conv_wgrad_partial")
                first += k

        if _launch_hook is not None:
            _launch_hook("conv_wgrad", shape, launch)
        else:
            launch()
        check(lib().mas_wgrad_reduce(_ptr(ws), C.c_void_p(pb0) if want_bias else None, tot, _ptr(dwo), _ptr(db), cout, cin, ks, _stream()),
              "wgrad_reduce")
        return dwo, db
    # (Round 6: mas_conv_wgrad_splits is > 0 for every geometry ops produces -- the general kernels have a slab mode --, so this route is
    #  only reached with MAS_WGRAD_GENERAL_SLABS=0.)  The other split-K kernels ADD into a zero accumulator.  One persistent scratch per (device, stream), zeroed once, is handed to every
    # weight-gradient launch; ``mas_wgrad_commit`` moves the sums into a fresh OIHW gradient tensor (+ bias gradient) and zeroes the
    # scratch again while it reads it: no fill launch and no permute copy per convolution (round 2: 356 fills per VQ-IMG step).
    acc = _wgrad_scratch.get(key)
    if acc is None or acc.numel() < nw + cout:
        acc = _wgrad_scratch[key] = torch.zeros(max(nw + cout, 1 << 22), dtype=torch.float32, device=x.device)

    def launch():
        for (n0, n1), d in zip(slices, descs):
            check(lib().mas_conv_wgrad(C.byref(d), _ptr(x[n0:n1]), _ptr(ss[n0:n1] if ss is not None else None), _ptr(dy[n0:n1]), _ptr(acc),
                                       C.c_void_p(acc.data_ptr() + 4 * nw) if want_bias else None, _stream()), "conv_wgrad")

    try:
        if _launch_hook is not None:
            _launch_hook("conv_wgrad", shape, launch)
        else:
            launch()
        check(lib().mas_wgrad_commit(_ptr(acc), _ptr(dwo), _ptr(db), cout, cin, ks, _stream()), "wgrad_commit")
    except Exception:
        _wgrad_scratch.pop(key, None)               # its all-zero invariant can no longer be assumed
        raise
    return dwo, db


_wgrad_scratch = {}
_CONV1X1 = os.environ.get("MAS_CONV1X1", "1") == "1"                 # (read by the library too: conv1x1.hip)
_wgrad_partials = {}


def upsample2x(x):
    n, c, h, w = x.shape
    y = _empty_nhwc(n, c, 2 * h, 2 * w, x.dtype, x.device)
    check(lib().mas_upsample2x(_ptr(x), _ptr(y), _DT[x.dtype], n, h, w, c, _stream()), "upsample2x")
    return y


def sumpool2x(x):
    n, c, h2, w2 = x.shape
    y = _empty_nhwc(n, c, h2 // 2, w2 // 2, x.dtype, x.device)
    check(lib().mas_sumpool2x(_ptr(x), _ptr(y), _DT[x.dtype], n, h2 // 2, w2 // 2, c, _stream()), "sumpool2x")
    return y


def space_to_depth2x(x, ho, wo, pad):
    """[N,C,H,W] -> [N,4C,ho,wo]: channel (dy*2+dx)*C + c of output pixel (h',w') = x[c, 2h'+dy-pad, 2w'+dx-pad] (0 outside)."""
    n, c, h, w = x.shape
    y = _empty_nhwc(n, 4 * c, ho, wo, x.dtype, x.device)
    check(lib().mas_space_to_depth2x(_ptr(x), _ptr(y), _DT[x.dtype], n, h, w, c, ho, wo, int(pad), _stream()), "space_to_depth2x")
    return y


def zero_stuff2x(x, hout, wout):
    n, c, h, w = x.shape
    y = _empty_nhwc(n, c, hout, wout, x.dtype, x.device)
    check(lib().mas_zero_stuff2x(_ptr(x), _ptr(y), _DT[x.dtype], n, h, w, c, hout, wout, _stream()), "zero_stuff2x")
    return y


# --------------------------------------------------------------------------- #
# fused [GroupNorm (+SiLU)] -> conv (+bias, +residual)
# --------------------------------------------------------------------------- #
# Weight gradients on a SECOND stream (MAS_WGRAD_STREAM, default 1 since late round 6; 0 = everything on one stream).  The convolution
# kernels run into the package power cap while the GroupNorm passes that follow each data gradient stay 13 % under it
# (profiles/r05_energy_budget.txt); a layer's weight gradient depends on nothing its GroupNorm backward produces, so it can run BESIDE those
# passes instead of in front of the data gradient: launched on the side stream right after the data gradient has been issued (ordered behind
# it), joined before the autograd node returns -- every tensor crosses streams inside one node only, so the caching allocator needs no
# record_stream, and whoever consumes the node's outputs (autograd, DDP / GradReducer hooks, the optimizer) sees them ordered on the
# current stream as before.  The weight-gradient grid is sized for three quarters of the CUs in this mode (MAS_WGRAD_CUS=-1, set below
# unless the user set it): with a persistent work-group on every CU the tiny finalize launch between the two GroupNorm passes is not placed
# until the weight gradient retires.  Same kernels, same arithmetic; the split-K count follows the grid, so gradients differ from the
# one-stream form in summation order only (bitwise reproducible run to run either way).  Step 52.9 -> 51.55 ms on one box
# (profiles/r06_wgrad_stream.txt).  No CU masks (round 3's masked form lost 20 %).
_WGRAD_STREAM = os.environ.get("MAS_WGRAD_STREAM", "1") == "1"
_WGRAD_CUS_IS_OURS = _WGRAD_STREAM and "MAS_WGRAD_CUS" not in os.environ
if _WGRAD_CUS_IS_OURS:
    os.environ["MAS_WGRAD_CUS"] = "-1"                    # (libmas_hip.so reads it at every weight-gradient call)
# layers below this many input elements keep their weight gradient on the current stream (0: every layer.  Measured at 0 / 2^23 / 2^25 /
# 2^26: 50.36 / 50.66 / 51.05 / 51.23 ms -- the overlap pays even at the 16 x 16 maps; profiles/r06_wgrad_stream.txt)
_WGRAD_STREAM_MIN_ELEMS = int(os.environ.get("MAS_WGRAD_STREAM_MIN_ELEMS", "0"))
_side_streams = {}
_side_ok = {}


def _streams_overlap(main, side) -> bool:
    """One-time probe (about 1 ms): do kernels on ``side`` run BESIDE kernels on ``main``?  Not when the two streams share a HIP hardware
    queue (GPU_MAX_HW_QUEUES, default 4, handed out in creation order -- RCCL's streams usually hold the other three by the time the first
    backward runs; see mas_hip/__init__.py).  Two spin kernels, one per stream: overlapped they take one kernel's time, serialised two."""
    cycles = 1_000_000
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    torch.cuda._sleep(1000)                                  # (first launches: code object load, the side stream's queue is made at first use)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.cuda._sleep(1000)
    main.wait_stream(side)
    ev[0].record(main)
    torch.cuda._sleep(cycles)
    ev[1].record(main)                                       # alone: ev0 -> ev1
    side.wait_stream(main)
    ev[2].record(main)
    torch.cuda._sleep(cycles)
    with torch.cuda.stream(side):
        torch.cuda._sleep(cycles)
        ev[3].record(side)
    ev[4].record(main)
    main.wait_stream(side)
    ev[3].synchronize(), ev[4].synchronize()
    alone = ev[0].elapsed_time(ev[1])
    both = max(ev[2].elapsed_time(ev[3]), ev[2].elapsed_time(ev[4]))
    _streams_overlap.last = (alone, both)
    return both < 1.6 * alone


def _side_stream_on() -> bool:
    """MAS_WGRAD_STREAM=1 and the side stream really runs beside the current one (probed once per device, at the first backward)"""
    if not _WGRAD_STREAM:
        return False
    dev = torch.cuda.current_device()
    ok = _side_ok.get(dev)
    if ok is None and torch.cuda.is_current_stream_capturing():
        return False                                           # (no probe inside a graph capture; asked again at the next eager backward)
    if ok is None:
        import warnings
        try:
            ok = os.environ.get("MAS_WGRAD_STREAM_PROBE", "1") != "1" or _streams_overlap(torch.cuda.current_stream(), _side_stream())
        except Exception as e:                                  # (torch.cuda._sleep is a private helper: without it, one stream)
            warnings.warn(f"mas_hip: the side-stream probe could not run ({type(e).__name__}: {e}); weight gradients stay on the current stream")
            _streams_overlap.last = (float("nan"), float("nan"))
            ok = False
        _side_ok[dev] = ok
        if not ok and _WGRAD_CUS_IS_OURS:
            os.environ["MAS_WGRAD_CUS"] = "0"               # the 3/4 grid only pays beside the GroupNorm passes
        if not ok and _streams_overlap.last[0] == _streams_overlap.last[0]:        # (not after a probe that could not run: it has warned already)
            warnings.warn("mas_hip: the weight-gradient side stream shares a HIP hardware queue with the current stream (its kernels would "
                          "serialise: one spin kernel %.3f ms, one per stream %.3f ms): weight gradients stay on the current stream.  Export GPU_MAX_HW_QUEUES=8 (or import mas_hip before "
                          "the first torch.cuda call) to get the overlapped schedule." % _streams_overlap.last)
    return ok


def _side_stream():
    dev = torch.cuda.current_device()
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def _on_side_stream(fn):
    """fn()'s launches go to the side stream, ordered behind everything issued on the current stream so far"""
    side = _side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        return fn()


def _join_side_stream():
    torch.cuda.current_stream().wait_stream(_side_stream())


class _NormActConv(torch.autograd.Function):
    """y = conv(act(gn(x)), W) + b (+ residual).   act in {none, affine, affine+silu}.

    Forward: gn statistics kernel (if act) -> conv kernel with the normalise/affine/SiLU fused
    into its loader.  Backward: wgrad kernel (recomputes the activated input in its loader),
    data gradient = the forward kernel on dy with flipped/transposed weights, then the
    GroupNorm(+SiLU) backward kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, gn_w, gn_b, residual, cfg, xpart=None, xrows=0):
        _require_cuda(x, "conv")
        cd = cfg["in_dtype"]
        x = nhwc(x, cd)
        n, cin, h, w = x.shape
        cout, cin_w, ks, _ = weight.shape
        if cin_w != cin:
            raise RuntimeError(f"conv: input has {cin} channels, weight expects {cin_w}")
        stride, pt, pl, pb, pr_, ups = cfg["stride"], cfg["pad_top"], cfg["pad_left"], cfg["pad_bottom"], cfg["pad_right"], cfg["upsample"]
        hl, wl = (2 * h, 2 * w) if ups else (h, w)
        ho = (hl + pt + pb - ks) // stride + 1
        wo = (wl + pl + pr_ - ks) // stride + 1
        act = cfg["act"]
        mr = ss = a = None
        pointwise = (ks == 1 and stride == 1 and not ups and pt == 0 and pl == 0 and _CONV1X1 and cin % 64 == 0 and cout % 128 == 0
                     and x.dtype == torch.bfloat16 and cfg["out_dtype"] == torch.bfloat16)
        mat = act != ACT_NONE and _MATERIALIZE and ((ks == 3 and not ups) or pointwise) and _gn_act_ok(cin, x.dtype)
        if act != ACT_NONE:
            if mat and xpart is None and gn_small_ok(x, cfg["groups"]):
                # small map, no statistics from the producer: statistics + activation in one launch (three otherwise)
                mr, ss, a = gn_stats_act(x, gn_w.detach().float(), gn_b.detach().float(), cfg["groups"], cfg["eps"], act)
            else:
                mr, ss = gn_stats(x, gn_w.detach().float(), gn_b.detach().float(), cfg["groups"], cfg["eps"], xpart, xrows)
        ctx.w_sources = getattr(weight, "_mas_sources", None)      # (the saved tensor comes back as another Python object)
        wp = ConvWeight(weight, False, ctx.w_sources)
        b32 = bias.detach().float() if bias is not None else None
        res = nhwc(residual, cd) if residual is not None else None
        # (needs_input_grad reflects requires_grad of the inputs even under torch.no_grad(); cfg["grad"] is the caller's grad mode)
        need_wgrad = cfg["grad"] and (ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]))
        # (1x1: the GEMM kernel (conv1x1.hip) has no prologue; one small pass + GEMM beats conv_fwd.hip's fused KS = 1 instance with or
        #  without a weight gradient to share the tensor with: 512 -> 1536 @16^2: 6 + 22 us against 52)
        if mat:
            # the activation as a tensor: this convolution (and, with a weight gradient, that too) runs prologue-free on it.  Also
            # without a second consumer (torch.no_grad(), frozen weights): 0.20 + 0.50 ms beats the fused loader's 0.73 ms at
            # 128 ch @256^2 (0.20 + 0.60 against 0.93 with the residual epilogue), 6 + 51 us against 74 us at 512 ch @16^2
            if a is None:
                a = gn_act(x, ss, act)
            y, ypart, yrows = conv_fwd_raw(a, None, wp, b32, res, n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, ACT_NONE, ups,
                                           cfg["out_dtype"], want_stats=True)
            if not need_wgrad or not _save_act["on"]:
                a = None                            # nothing in the backward reads it (or MAS_SAVE_ACT=0: recomputed there): not saved
        else:
            y, ypart, yrows = conv_fwd_raw(x, ss, wp, b32, res, n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, act, ups,
                                           cfg["out_dtype"], want_stats=True)
        _stats_state["stash"] = (ypart, yrows) if ypart is not None else None
        ctx.cfg = cfg
        ctx.dims = (n, h, w, cin, ho, wo, cout, ks)
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, gn_w, mr, ss, a)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, gn_w, mr, ss, a = ctx.saved_tensors
        cfg = ctx.cfg
        n, h, w, cin, ho, wo, cout, ks = ctx.dims
        cd = cfg["in_dtype"]
        stride, pt, pl, ups, act = cfg["stride"], cfg["pad_top"], cfg["pad_left"], cfg["upsample"], cfg["act"]
        dy = nhwc(dy, cd)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        need_gn = act != ACT_NONE and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        dx = dw = db = dgw = dgb = None

        def wgrad():
            if a is not None:          # the forward left the activated input: prologue-free weight gradient
                return conv_wgrad_raw(a, None, dy, n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, ACT_NONE, ups, need_b)
            return conv_wgrad_raw(x, ss, dy, n, h, w, cin, ho, wo, cout, ks, stride, pt, pl, act, ups, need_b)

        # (MAS_WGRAD_STREAM: beside the GroupNorm backward, behind the data gradient -- see _on_side_stream)
        defer = (need_w or need_b) and act != ACT_NONE and (need_x or need_gn) and dy.is_cuda and x.numel() >= _WGRAD_STREAM_MIN_ELEMS and _side_stream_on()
        if (need_w or need_b) and not defer:
            dw, db = wgrad()
            dw = dw.to(weight.dtype) if need_w else None
        if need_x or need_gn:
            wt = ConvWeight(weight, True, ctx.w_sources)
            hl, wl = (2 * h, 2 * w) if ups else (h, w)
            da = None
            if stride == 1:
                d_in, hd, wd = dy, ho, wo
            else:
                esz = dy.element_size()
                slices = _batch_slices(n, h * w * cin * esz, ho * wo * cout * esz)          # (tensors of 2^31 bytes and more: conv_fwd_raw)
                dfws = [_desc(n1 - n0, h, w, cin, ho, wo, cout, ks, stride, pt, pl, cd, cd, ACT_NONE, ups) for n0, n1 in slices]
                if ks == 3 and stride == 2 and all(lib().mas_conv_s2_dgrad_supported(C.byref(dfw)) for dfw in dfws):
                    # Downsample: the four parity classes of dx straight from dy (conv_s2.hip), exact FLOPs, no zero-stuffed tensor
                    wpk = _pack_cache.get(weight, True, cd, WLAYOUT_K64, ctx.w_sources)
                    da = torch.empty((n, cin, h, w), dtype=cd, device=dy.device, memory_format=torch.channels_last)
                    for (n0, n1), dfw in zip(slices, dfws):
                        check(lib().mas_conv_s2_dgrad(C.byref(dfw), _ptr(dy[n0:n1]), _ptr(wpk), _ptr(da[n0:n1]), _stream()), "conv_s2_dgrad")
                else:  # adjoint of the strided read: zero-stuff dy, then a stride-1 conv
                    hd, wd = (ho - 1) * stride + 1, (wo - 1) * stride + 1
                    d_in = zero_stuff2x(dy, hd, wd)
            if da is None and ups and stride == 1:
                # Upsample + conv: the data gradient with respect to the LOW-resolution input straight from dy (conv_up2.hip: the four phase
                # images of dy through the transposed 2x2 phase weights) -- no high-resolution da, no sum-pooling pass
                esz = dy.element_size()
                slices = _batch_slices(n, h * w * cin * esz, ho * wo * cout * esz)
                dfws = [_desc(n1 - n0, h, w, cin, ho, wo, cout, ks, 1, pt, pl, cd, cd, ACT_NONE, True) for n0, n1 in slices]
                if ks == 3 and all(_up2_supported(dfw, True) for dfw in dfws):
                    wpk = _pack_cache.get(weight, True, cd, WLAYOUT_UP2, ctx.w_sources)
                    da = torch.empty((n, cin, h, w), dtype=cd, device=dy.device, memory_format=torch.channels_last)

                    def launch_up2():
                        for (n0, n1), dfw in zip(slices, dfws):
                            check(lib().mas_conv_up2_dgrad(C.byref(dfw), _ptr(dy[n0:n1]), _ptr(wpk), _ptr(da[n0:n1]), _stream()), "conv_up2_dgrad")

                    if _launch_hook is not None:
                        _launch_hook("conv_up2_dgrad", (n, h, w, cin, ho, wo, cout, ks, 1, ACT_NONE, 0), launch_up2)
                    else:
                        launch_up2()
            if da is None:
                da = conv_fwd_raw(d_in, None, wt, None, None, n, hd, wd, cout, hl, wl, cin, ks, 1, ks - 1 - pt, ks - 1 - pl, ACT_NONE, False, cd)
                if ups:
                    da = sumpool2x(da)
            if defer:
                dw, db = _on_side_stream(wgrad)
            if act != ACT_NONE:
                dx, dgw, dgb = gn_bwd(x, da, None, cfg["groups"], act, gn_w.detach().float(), mr, ss)
                dgw, dgb = dgw.to(gn_w.dtype), dgb.to(gn_w.dtype)
            else:
                dx = da
            if defer:
                _join_side_stream()
                dw = dw.to(weight.dtype) if need_w else None
        dres = dy if ctx.has_res and ctx.needs_input_grad[5] else None
        return dx, dw, (db.to(weight.dtype) if db is not None else None), dgw, dgb, dres, None, None, None


def norm_act_conv(x, weight, bias, gn_w=None, gn_b=None, residual=None, *, stride=1, padding=(1, 1, 1, 1), act=ACT_NONE,
                  upsample=False, groups=32, eps=1e-6, in_dtype=None, out_dtype=None):
    """padding = (top, bottom, left, right)."""
    cd = in_dtype or compute_dtype()
    cfg = dict(stride=stride, pad_top=padding[0], pad_bottom=padding[1], pad_left=padding[2], pad_right=padding[3], act=act,
               upsample=bool(upsample), groups=groups, eps=eps, in_dtype=cd, out_dtype=out_dtype or cd, grad=torch.is_grad_enabled())
    if cfg["in_dtype"] == torch.float32 and cfg["out_dtype"] != torch.float32:
        raise RuntimeError("conv: fp32 input requires fp32 output")
    # Channel counts that are not a multiple of one 16-byte slot (RGB in/out, the 159 VQ-SEG classes) would take the
    # kernels' element-wise loaders; zero-padding the channel axis (differentiable torch ops on tiny tensors; the padded
    # weights are exactly zero-gradient-free slices) keeps every launch on the vectorised / transpose-read paths.
    epu = 8 if cd == torch.bfloat16 else 4
    cout, cin = weight.shape[0], weight.shape[1]
    if act == ACT_NONE and cin % epu:
        padc = epu - cin % epu
        x = torch.nn.functional.pad(nhwc(x, cd), (0, 0, 0, 0, 0, padc))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, padc))
    if cout % epu and residual is None:
        padc = epu - cout % epu
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, padc))
        if bias is not None:
            bias = torch.nn.functional.pad(bias, (0, padc))
        return _NormActConv.apply(x, weight, bias, gn_w, gn_b, residual, cfg)[:, :cout]
    xpart, xrows = _take_stats(x) if act != ACT_NONE else (None, 0)
    return _attach_stats(_NormActConv.apply(x, weight, bias, gn_w, gn_b, residual, cfg, xpart, xrows))


class _ResBlock(torch.autograd.Function):
    """shortcut(x) + conv2(silu(gn2(conv1(silu(gn1(x))))))  (reference ResnetBlock.forward, models/modules.py:119-136) as ONE autograd
    node; shortcut = identity for Cin == Cout, the block's 1x1 ``nin_shortcut`` otherwise (round 4).  Same kernels as two
    ``_NormActConv`` calls (+ the 1x1), but the backward hands the skip-connection gradient -- dy itself, or the 1x1's data gradient
    of dy -- to the GroupNorm-backward kernel of norm1 (``dres``), so the two gradient branches of x are summed inside that
    streaming pass instead of by a separate elementwise add over the full activation."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, c1w, c1b, n2w, n2b, c2w, c2b, sw, sb, groups, eps, cd, xpart=None, xrows=0, grad=True):
        _require_cuda(x, "resblock")
        x = nhwc(x, cd)
        n, c, h, w = x.shape
        co = c1w.shape[0]
        f32 = lambda t: t.detach().float() if t is not None else None
        ng = ctx.needs_input_grad
        mat = _MATERIALIZE and _gn_act_ok(c, x.dtype) and _gn_act_ok(co, x.dtype)

        def norm(inp, gw, gb, part, rows):
            """-> (mean_rstd, scale_shift, activated tensor or None): small maps without statistics from the producer take ONE launch"""
            if mat and part is None and gn_small_ok(inp, groups):
                return gn_stats_act(inp, f32(gw), f32(gb), groups, eps, ACT_AFFINE_SILU)
            return gn_stats(inp, f32(gw), f32(gb), groups, eps, part, rows) + (None,)

        def conv(inp, ss_, a_, wgt, bia, resid, need_w, ci):
            """-> (output, statistics table, rows, activated input if the backward wants it)"""
            if mat:                                  # (with or without a weight gradient to share it with: see _NormActConv.forward)
                if a_ is None:
                    a_ = gn_act(inp, ss_, ACT_AFFINE_SILU)
                return conv_fwd_raw(a_, None, ConvWeight(wgt, False), f32(bia), resid, n, h, w, ci, h, w, co, 3, 1, 1, 1, ACT_NONE, False, cd,
                                    want_stats=True) + (a_ if grad and need_w and _save_act["on"] else None,)
            return conv_fwd_raw(inp, ss_, ConvWeight(wgt, False), f32(bia), resid, n, h, w, ci, h, w, co, 3, 1, 1, 1, ACT_AFFINE_SILU, False,
                                cd, want_stats=True) + (None,)

        mr1, ss1, act1 = norm(x, n1w, n1b, xpart, xrows)
        hh, hpart, hrows, a1 = conv(x, ss1, act1, c1w, c1b, None, ng[3] or ng[4], c)
        # the skip path: x itself, or nin_shortcut(x) (a plain 1x1 on the block's INPUT, models/modules.py:131-134)
        skip = x if sw is None else conv_fwd_raw(x, None, ConvWeight(sw, False), f32(sb), None, n, h, w, c, h, w, co, 1, 1, 0, 0, ACT_NONE, False, cd)
        mr2, ss2, act2 = norm(hh, n2w, n2b, hpart, hrows)
        y, ypart, yrows, a2 = conv(hh, ss2, act2, c2w, c2b, skip, ng[7] or ng[8], co)
        _stats_state["stash"] = (ypart, yrows) if ypart is not None else None
        ctx.groups, ctx.cd, ctx.has_sc = groups, cd, sw is not None
        ctx.save_for_backward(x, hh, mr1, ss1, mr2, ss2, n1w, c1w, n2w, c2w, a1, a2, sw)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, hh, mr1, ss1, mr2, ss2, n1w, c1w, n2w, c2w, a1, a2, sw = ctx.saved_tensors
        cd, groups = ctx.cd, ctx.groups
        n, c, h, w = x.shape
        co = c1w.shape[0]
        dy = nhwc(dy, cd)
        ng = ctx.needs_input_grad
        geo1 = (n, h, w, c, h, w, co, 3, 1, 1, 1)
        geo2 = (n, h, w, co, h, w, co, 3, 1, 1, 1)
        geo1t = (n, h, w, co, h, w, c, 3, 1, 1, 1)           # conv1's data gradient: a convolution from Cout back to Cin
        dw2 = db2 = dw1 = db1 = dsw = dsb = None
        dx = dg1w = dg1b = None

        def wgrad2():
            return conv_wgrad_raw(a2, None, dy, *geo2, ACT_NONE, False, True) if a2 is not None else \
                conv_wgrad_raw(hh, ss2, dy, *geo2, ACT_AFFINE_SILU, False, True)

        def wgrad1(dh_):
            return conv_wgrad_raw(a1, None, dh_, *geo1, ACT_NONE, False, True) if a1 is not None else \
                conv_wgrad_raw(x, ss1, dh_, *geo1, ACT_AFFINE_SILU, False, True)

        need_x = ng[0] or ng[1] or ng[2]
        side = dy.is_cuda and x.numel() >= _WGRAD_STREAM_MIN_ELEMS and _side_stream_on()           # weight gradients beside the GroupNorm backward passes (see _on_side_stream)
        # conv2 / norm2
        if (ng[7] or ng[8]) and not side:       # (a2 / a1: the activated inputs the forward left behind -> prologue-free weight gradients)
            dw2, db2 = wgrad2()
        da2 = conv_fwd_raw(dy, None, ConvWeight(c2w, True), None, None, *geo2, ACT_NONE, False, cd)
        if (ng[7] or ng[8]) and side:
            dw2, db2 = _on_side_stream(wgrad2)
        dh, dg2w, dg2b = gn_bwd(hh, da2, None, groups, ACT_AFFINE_SILU, n2w.detach().float(), mr2, ss2)
        # the skip path's parameter gradients and its gradient with respect to x
        dskip = dy
        if ctx.has_sc:
            if ng[9] or ng[10]:
                dsw, dsb = conv_wgrad_raw(x, None, dy, n, h, w, c, h, w, co, 1, 1, 0, 0, ACT_NONE, False, True)
            if need_x:
                dskip = conv_fwd_raw(dy, None, ConvWeight(sw, True), None, None, n, h, w, co, h, w, c, 1, 1, 0, 0, ACT_NONE, False, cd)
        # conv1 / norm1 (+ the skip connection's gradient, fused into the GroupNorm-backward apply pass)
        if (ng[3] or ng[4]) and not (side and need_x):
            dw1, db1 = wgrad1(dh) if not side else _on_side_stream(lambda: wgrad1(dh))
        if need_x:
            da1 = conv_fwd_raw(dh, None, ConvWeight(c1w, True), None, None, *geo1t, ACT_NONE, False, cd)
            if (ng[3] or ng[4]) and side:
                dw1, db1 = _on_side_stream(lambda: wgrad1(dh))
            dx, dg1w, dg1b = gn_bwd(x, da1, dskip, groups, ACT_AFFINE_SILU, n1w.detach().float(), mr1, ss1)
        if side:
            _join_side_stream()
        cast = lambda g, ref: g.to(ref.dtype) if g is not None else None
        return (dx, cast(dg1w, n1w), cast(dg1b, n1w), cast(dw1, c1w), cast(db1, c1w), cast(dg2w, n2w), cast(dg2b, n2w),
                cast(dw2, c2w), cast(db2, c2w), cast(dsw, sw) if sw is not None else None, cast(dsb, sw) if sw is not None else None,
                None, None, None, None, None, None)


def resblock(x, norm1, conv1, norm2, conv2, shortcut=None):
    """``shortcut``: the block's ``nin_shortcut`` module (1x1, stride 1, no padding) when Cin != Cout, else None"""
    xpart, xrows = _take_stats(x)
    sw, sb = (shortcut.weight, shortcut.bias) if shortcut is not None else (None, None)
    return _attach_stats(_ResBlock.apply(x, norm1.weight, norm1.bias, conv1.weight, conv1.bias, norm2.weight, norm2.bias, conv2.weight,
                                         conv2.bias, sw, sb, norm1.num_groups, norm1.eps, compute_dtype(), xpart, xrows, torch.is_grad_enabled()))


# --------------------------------------------------------------------------- #
# vector quantiser
# --------------------------------------------------------------------------- #
class _VQ(torch.autograd.Function):
    """(z_q straight-through, loss, idx) = VQ(z, codebook)  -- Codebook.forward's arithmetic
    (reference models/modules.py:501-517)."""

    @staticmethod
    def forward(ctx, z, codebook, beta):
        _require_cuda(z, "vq")
        z = nhwc(z, torch.float32)
        n, d, h, w = z.shape
        k = codebook.shape[0]
        m = n * h * w
        cb = codebook.detach().contiguous().float()
        idx = torch.empty(m, dtype=torch.int64, device=z.device)
        zq = torch.empty((n, d, h, w), dtype=torch.float32, device=z.device, memory_format=torch.channels_last)
        sq = torch.empty(1, dtype=torch.float32, device=z.device)
        wsb = lib().mas_vq_workspace(m, k)
        ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=z.device)
        check(lib().mas_vq_argmin_fwd(_ptr(z), _ptr(cb), m, k, d, _ptr(idx), _ptr(zq), _ptr(sq), _ptr(ws), wsb, _stream()), "vq_argmin_fwd")
        loss = (sq[0] * ((1.0 + beta) / (m * d)))
        ctx.beta = beta
        ctx.save_for_backward(z, cb, idx)
        ctx.mark_non_differentiable(idx)
        return zq, loss, idx

    @staticmethod
    def backward(ctx, g_zq, g_loss, _g_idx):
        z, cb, idx = ctx.saved_tensors
        n, d, h, w = z.shape
        m, k = n * h * w, cb.shape[0]
        need_z, need_cb = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_zq = nhwc(g_zq, torch.float32) if g_zq is not None else None
        g_loss = g_loss.reshape(1).float().contiguous() if g_loss is not None else None
        dz = torch.empty_like(z, memory_format=torch.channels_last) if need_z else None
        dcb = torch.zeros_like(cb) if need_cb else None
        check(lib().mas_vq_bwd(_ptr(z), _ptr(cb), _ptr(idx), _ptr(g_zq), _ptr(g_loss), float(ctx.beta), m, k, d, _ptr(dz), _ptr(dcb),
                               _stream()), "vq_bwd")
        return dz, dcb, None


_VQ_DIMS = (32, 64, 128, 256)


def vq_lookup(z, codebook, beta):
    """``Codebook.forward``'s lookup (reference models/modules.py:501-517).  The kernel takes codebook_dim in {32, 64, 128, 256}; any other
    width up to 256 is zero-padded to the next of those with differentiable torch ops (zero dimensions add nothing to any distance, to the
    loss's sum of squares or to a gradient; the loss is re-normalised to the true width), so every ``embed_dim`` the reference's constructor
    accepts up to 256 works -- found by tests/test_gpu_model.py's off-config case (round 6)."""
    d = codebook.shape[1]
    if d in _VQ_DIMS:
        return _VQ.apply(z, codebook, beta)
    if d > _VQ_DIMS[-1]:
        # wider than any kernel instantiation: the reference's own formula (models/modules.py:501-517) from ATen ops on the GPU
        zp = nhwc(z, torch.float32).permute(0, 2, 3, 1)
        zf = zp.reshape(-1, d)
        cb = codebook.float()
        dist = (zf * zf).sum(1, keepdim=True) + (cb * cb).sum(1) - 2.0 * zf @ cb.t()
        idx = torch.argmin(dist, dim=1)
        zq = torch.nn.functional.embedding(idx, cb).view(zp.shape)
        loss = torch.mean((zq.detach() - zp) ** 2) + beta * torch.mean((zq - zp.detach()) ** 2)
        return (zp + (zq - zp).detach()).permute(0, 3, 1, 2), loss, idx
    dp = next(v for v in _VQ_DIMS if v >= d)
    zp = torch.nn.functional.pad(nhwc(z, torch.float32), (0, 0, 0, 0, 0, dp - d))
    cp = torch.nn.functional.pad(codebook.float(), (0, dp - d))
    zq, loss, idx = _VQ.apply(zp, cp, beta)
    return zq[:, :d], loss * (dp / d), idx


# --------------------------------------------------------------------------- #
# SyncBatchNorm behind quant_conv (reference models/vqvae.py:15-16)
# --------------------------------------------------------------------------- #
def _bn_sums(x2d, dy2d, mean_rstd):
    m, c = x2d.shape
    sums = torch.empty(2 * c + 1, dtype=torch.float64, device=x2d.device)
    wsb = lib().mas_bn_workspace(m, c)
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=x2d.device)
    check(lib().mas_bn_partial_sums(_ptr(x2d), _ptr(dy2d), _ptr(mean_rstd), m, c, _ptr(sums), _ptr(ws), wsb, _stream()), "bn_partial_sums")
    return sums


class _SyncBatchNorm(torch.autograd.Function):
    """``torch.nn.SyncBatchNorm`` in training mode on an fp32 NHWC activation: per-rank sums (``mas_bn_partial_sums``, fixed order), ONE
    all_reduce of the fp64 vector {sum, sum of squares, count} over ``group`` when it has more than one rank (torch's own module
    gathers mean / invstd / count instead: the same statistics), then ``mas_bn_finalize`` (mean, rstd, the affine pair, running
    statistics) and ``mas_bn_apply``; the backward exchanges {sum dy, sum dy * xhat} the same way and returns the LOCAL sums as
    the weight / bias gradients, as torch does (data parallelism averages them afterwards)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running, eps, momentum, group):
        running_mean, running_var = running                 # (buffers, updated in place: handed over in a tuple, outside autograd's view)
        n, c, h, w = x.shape
        x = nhwc(x, torch.float32)
        x2 = x.permute(0, 2, 3, 1).reshape(-1, c)
        sums = _bn_sums(x2, None, None)
        if group is not None:
            torch.distributed.all_reduce(sums, group=group)
        mean_rstd = torch.empty((c, 2), dtype=torch.float32, device=x.device)
        ss = torch.empty((c, 2), dtype=torch.float32, device=x.device)
        w32 = weight.detach().float().contiguous() if weight is not None else None
        b32 = bias.detach().float().contiguous() if bias is not None else None
        check(lib().mas_bn_finalize(_ptr(sums), _ptr(w32), _ptr(b32), float(eps), float(momentum), _ptr(running_mean), _ptr(running_var),
                                    _ptr(mean_rstd), _ptr(ss), c, _stream()), "bn_finalize")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        check(lib().mas_bn_apply(_ptr(x), _ptr(ss), _ptr(y), n * h * w, c, _stream()), "bn_apply")
        ctx.save_for_backward(x, w32 if w32 is not None else torch.empty(0, device=x.device), mean_rstd, sums[2 * c:].clone())
        ctx.group, ctx.has_w, ctx.has_b = group, weight is not None, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, mean_rstd, count = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = nhwc(dy, torch.float32)
        sums = _bn_sums(x.permute(0, 2, 3, 1).reshape(-1, c), dy.permute(0, 2, 3, 1).reshape(-1, c), mean_rstd)
        db = sums[:c].float() if ctx.has_b else None            # this rank's sums: the parameter gradients
        dg = sums[c:2 * c].float() if ctx.has_w else None
        if ctx.group is not None:
            torch.distributed.all_reduce(sums, group=ctx.group)
            sums[2 * c:] = count                                # (= what the all_reduce summed; kept explicit: the forward's global count)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        check(lib().mas_bn_bwd_apply(_ptr(x), _ptr(dy), _ptr(mean_rstd), _ptr(w32) if ctx.has_w else None, _ptr(sums), _ptr(dx), n * h * w, c,
                                     _stream()), "bn_bwd_apply")
        return dx, dg, db, None, None, None, None


class _BatchNormEval(torch.autograd.Function):
    """``nn.SyncBatchNorm`` in evaluation mode: y = x * scale + shift with the pair from the RUNNING statistics (``mas_bn_finalize`` with no
    sums) + ``mas_bn_apply``.  Differentiable like ``F.batch_norm(training=False)`` (fine-tuning with frozen statistics, input
    gradients in eval()): dx = dy * gamma * rstd (``mas_bn_apply`` again, with a zero shift), dgamma = sum dy * xhat and dbeta = sum dy
    (``mas_bn_partial_sums`` against the running mean / rstd, fixed order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running, eps):
        running_mean, running_var = running
        n, c, h, w = x.shape
        x = nhwc(x, torch.float32)
        ss = torch.empty((c, 2), dtype=torch.float32, device=x.device)
        mean_rstd = torch.empty((c, 2), dtype=torch.float32, device=x.device)
        w32 = weight.detach().float().contiguous() if weight is not None else None
        b32 = bias.detach().float().contiguous() if bias is not None else None
        check(lib().mas_bn_finalize(None, _ptr(w32), _ptr(b32), float(eps), 0.0, _ptr(running_mean), _ptr(running_var), _ptr(mean_rstd), _ptr(ss),
                                    c, _stream()), "bn_finalize")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        check(lib().mas_bn_apply(_ptr(x), _ptr(ss), _ptr(y), n * h * w, c, _stream()), "bn_apply")
        ctx.save_for_backward(x, ss, mean_rstd)
        ctx.has_w, ctx.has_b = weight is not None, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ss, mean_rstd = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = nhwc(dy, torch.float32)
        dx = dg = db = None
        if ctx.needs_input_grad[0]:
            ss_dx = ss.clone()
            ss_dx[:, 1] = 0.0
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            check(lib().mas_bn_apply(_ptr(dy), _ptr(ss_dx), _ptr(dx), n * h * w, c, _stream()), "bn_apply")
        if (ctx.has_w and ctx.needs_input_grad[1]) or (ctx.has_b and ctx.needs_input_grad[2]):
            sums = _bn_sums(x.permute(0, 2, 3, 1).reshape(-1, c), dy.permute(0, 2, 3, 1).reshape(-1, c), mean_rstd)
            db = sums[:c].float() if ctx.has_b else None
            dg = sums[c:2 * c].float() if ctx.has_w else None
        return dx, dg, db, None, None


def sync_batch_norm(x, weight, bias, running_mean, running_var, eps, momentum, training, group=None):
    """``nn.SyncBatchNorm.forward`` on a 4-D fp32 CUDA activation.  ``group``: a process group with more than one rank, or None (no
    exchange).  Evaluation: the affine pair from the running statistics (``mas_bn_finalize`` with no sums) + ``mas_bn_apply``, as an
    autograd node (``_BatchNormEval``): eval() with frozen statistics stays differentiable w.r.t. x, weight and bias as torch's does."""
    _require_cuda(x, "sync_batch_norm")
    if training:
        return _SyncBatchNorm.apply(x, weight, bias, (running_mean, running_var), eps, momentum, group)
    return _BatchNormEval.apply(x, weight, bias, (running_mean, running_var), eps)


# --------------------------------------------------------------------------- #
# single-head spatial attention core (AttnBlock, reference models/modules.py:174-187)
# --------------------------------------------------------------------------- #
class _SpatialAttention(torch.autograd.Function):
    """h = softmax_keys(q k^T C^-1/2) v over the h*w tokens of each image, on the fused qkv projection [N,3C,H,W] (channels_last):
    ``mas_spatial_attn_fwd / _bwd`` (one forward launch, two backward launches; the [S,S] scores live in LDS)."""

    @staticmethod
    def forward(ctx, qkv, c):
        n, c3, h, w = qkv.shape
        x = nhwc(qkv)                                               # memory = [N, H*W, 3C]
        o = _empty_nhwc(n, c, h, w, x.dtype, x.device)
        lse = torch.empty((n, h * w), dtype=torch.float32, device=x.device)
        check(lib().mas_spatial_attn_fwd(_ptr(x), _ptr(o), _ptr(lse), _DT[x.dtype], n, h * w, c, _stream()), "spatial_attn_fwd")
        ctx.c = c
        ctx.save_for_backward(x, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        x, lse = ctx.saved_tensors
        n, c3, h, w = x.shape
        g = nhwc(do, x.dtype)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        delta = torch.empty_like(lse)
        check(lib().mas_spatial_attn_bwd(_ptr(x), _ptr(g), _ptr(lse), _ptr(delta), _ptr(dx), _DT[x.dtype], n, h * w, ctx.c, _stream()),
              "spatial_attn_bwd")
        return dx, None


class _BatchNormLeaky(torch.autograd.Function):
    """``nn.BatchNorm2d`` (per-rank batch statistics in training, running statistics in evaluation) + the ``nn.LeakyReLU(slope)`` that
    follows it, on a bf16 or fp32 NHWC map in its own storage type: the normalisation pairs of the PatchGAN discriminator (reference
    losses/discriminator.py:26-33).  Forward: fixed-order fp64 sums (``mas_bn_partial_sums_act``), ``mas_bn_finalize`` (mean / rstd, the
    affine pair, running statistics), ``mas_bn_apply_act`` (one read, one write, the activation fused).  Backward: the sums of
    g = dy * lrelu'(u) -- u recomputed from x, nothing but x is saved -- are this layer's dbeta / dgamma; ``mas_bn_bwd_apply_act`` writes dx.
    Evaluation: dx = g * gamma * rstd (the same kernel with zero sums)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running, eps, momentum, slope, training):
        running_mean, running_var = running
        n, c, h, w = x.shape
        x = nhwc(x)
        dt = _DT[x.dtype]
        m = n * h * w
        x2 = x.permute(0, 2, 3, 1).reshape(-1, c)
        sums = None
        if training:
            sums = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)
            wsb = lib().mas_bn_workspace(m, c)
            ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=x.device)
            check(lib().mas_bn_partial_sums_act(_ptr(x2), None, None, None, 1.0, dt, m, c, _ptr(sums), _ptr(ws), wsb, _stream()), "bn_partial_sums")
        mean_rstd = torch.empty((c, 2), dtype=torch.float32, device=x.device)
        ss = torch.empty((c, 2), dtype=torch.float32, device=x.device)
        w32 = weight.detach().float().contiguous() if weight is not None else None
        b32 = bias.detach().float().contiguous() if bias is not None else None
        check(lib().mas_bn_finalize(_ptr(sums), _ptr(w32), _ptr(b32), float(eps), float(momentum), _ptr(running_mean), _ptr(running_var),
                                    _ptr(mean_rstd), _ptr(ss), c, _stream()), "bn_finalize")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        check(lib().mas_bn_apply_act(_ptr(x), _ptr(ss), _ptr(y), float(slope), dt, m, c, _stream()), "bn_apply")
        ctx.save_for_backward(x, w32 if w32 is not None else torch.empty(0, device=x.device), mean_rstd, ss)
        ctx.has_w, ctx.has_b, ctx.slope, ctx.training = weight is not None, bias is not None, float(slope), bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, mean_rstd, ss = ctx.saved_tensors
        n, c, h, w = x.shape
        m = n * h * w
        dt = _DT[x.dtype]
        dy = nhwc(dy, x.dtype)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)
        wsb = lib().mas_bn_workspace(m, c)
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=x.device)
        check(lib().mas_bn_partial_sums_act(_ptr(x), _ptr(dy), _ptr(mean_rstd), _ptr(ss), ctx.slope, dt, m, c, _ptr(sums), _ptr(ws), wsb, _stream()),
              "bn_partial_sums")
        db = sums[:c].float() if ctx.has_b else None
        dg = sums[c:2 * c].float() if ctx.has_w else None
        dx = None
        if ctx.needs_input_grad[0]:
            if not ctx.training:                                # frozen statistics: no mean terms
                sums = torch.zeros_like(sums)
                sums[2 * c] = 1.0
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            check(lib().mas_bn_bwd_apply_act(_ptr(x), _ptr(dy), _ptr(mean_rstd), _ptr(w32) if ctx.has_w else None, _ptr(ss), ctx.slope, _ptr(sums),
                                             _ptr(dx), dt, m, c, _stream()), "bn_bwd_apply")
        return dx, dg, db, None, None, None, None, None


def batch_norm_leaky_relu(x, bn: torch.nn.BatchNorm2d, slope: float):
    """``LeakyReLU(slope)(bn(x))`` for a 4-D bf16 / fp32 CUDA map on ``batchnorm.hip`` (``slope`` 1.0: the normalisation alone), with
    ``nn.BatchNorm2d``'s bookkeeping: ``num_batches_tracked``, cumulative momentum, running statistics, evaluation mode."""
    _require_cuda(x, "batch_norm_leaky_relu")
    training = bn.training or not bn.track_running_stats
    if training and x.numel() // x.shape[1] == 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {x.shape}")
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BatchNormLeaky.apply(x, bn.weight, bn.bias, (rm, rv), bn.eps, momentum, slope, training)


_spatial_attn_force = {"on": False}


def force_bf16_spatial_attention(on: bool) -> bool:
    """Tests: route fp32-mode AttnBlock cores through the bf16 kernels of ``spatial_attn.hip`` (inputs rounded to bf16, output cast
    back).  Returns the previous setting."""
    old = _spatial_attn_force["on"]
    _spatial_attn_force["on"] = bool(on)
    return old


def spatial_attention(qkv: torch.Tensor, c: int) -> torch.Tensor:
    """qkv: [N,3C,H,W] channels_last (q|k|v stacked on channels).  softmax over keys of q.k^T * C^-1/2, then . v
    (reference models/modules.py:174-187).  bf16 (the compute dtype of the benched path): the hand-written HIP kernels, for the
    shapes they cover (H*W <= 256 tokens, C <= 512, C % 32 == 0 -- every AttnBlock of the reference's configs).  fp32 (the parity /
    debug mode) and anything outside that envelope: two batched library GEMMs on views of the NHWC buffer + a row softmax."""
    n, c3, h, w = qkv.shape
    if qkv.is_cuda and qkv.dtype == torch.bfloat16 and h * w <= 256 and c <= 512 and c % 32 == 0:
        return _SpatialAttention.apply(qkv, c)
    if _spatial_attn_force["on"] and qkv.is_cuda and qkv.dtype == torch.float32 and h * w <= 256 and c <= 512 and c % 32 == 0:
        # parity knob (tests only): the fp32 MODE with its attention cores on the bf16 HIP kernel, so that the kernel is held against
        # the reference's fp32 golden inside an otherwise exact-fp32 model (VERDICT r5 next #3d)
        return _SpatialAttention.apply(qkv.to(torch.bfloat16), c).float()
    t = qkv.permute(0, 2, 3, 1).reshape(n, h * w, c3)          # a view of the NHWC buffer
    q, k, v = t[..., :c], t[..., c:2 * c], t[..., 2 * c:]
    s = torch.bmm(q, k.transpose(1, 2)) * (int(c) ** (-0.5))
    p = torch.softmax(s, dim=2)
    o = torch.bmm(p, v)
    return o.reshape(n, h, w, c).permute(0, 3, 1, 2)            # [N,C,H,W] logical, NHWC memory


# --------------------------------------------------------------------------- #
# causal multi-head attention (transformer path, reference models/transformer.py:44-115)
# --------------------------------------------------------------------------- #
_ATTN_HEAD_DIMS = (16, 32, 64, 128)


class _CausalAttention(torch.autograd.Function):
    """context = softmax_causal((q/sqrt(hd)) k^T) v on the fused qkv projection [B,S,3*H*hd].
    Forward and backward are the flash-style HIP kernels (the [S,S] scores are never written; the backward
    recomputes them tile by tile from the saved log-sum-exp and writes d(qkv) in place, no atomics)."""

    @staticmethod
    def forward(ctx, qkv, n_heads, cd, scale=None):
        _require_cuda(qkv, "causal_attention")
        x = qkv.to(cd).contiguous()
        b, s, d3 = x.shape
        d = d3 // 3
        hd = d // n_heads
        scale = float(hd) ** -0.5 if scale is None else float(scale)      # (a zero-padded head keeps the scale of its true width)
        o = torch.empty((b, s, d), dtype=cd, device=x.device)
        lse = torch.empty((b, n_heads, s), dtype=torch.float32, device=x.device)
        esz = x.element_size()
        base = x.data_ptr()
        check(lib().mas_attn_causal_fwd(C.c_void_p(base), C.c_void_p(base + d * esz), C.c_void_p(base + 2 * d * esz), _ptr(o), _ptr(lse),
                                        _DT[cd], b, n_heads, s, hd, d3, d3, d3, s * d3, s * d3, s * d3, scale, _stream()),
              "attn_causal_fwd")
        ctx.n_heads = n_heads
        ctx.scale = scale
        ctx.in_dtype = qkv.dtype
        ctx.save_for_backward(x, o, lse)
        return o.to(qkv.dtype)

    @staticmethod
    def backward(ctx, do):
        x, o, lse = ctx.saved_tensors
        b, s, d3 = x.shape
        h = ctx.n_heads
        hd = d3 // 3 // h
        g = do.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        delta = torch.empty_like(lse)
        check(lib().mas_attn_causal_bwd(_ptr(x), _ptr(o), _ptr(g), _ptr(lse), _ptr(delta), _ptr(dx), _DT[x.dtype], b, h, s, hd,
                                        ctx.scale, _stream()), "attn_causal_bwd")
        return dx.to(ctx.in_dtype), None, None, None


def causal_attention(qkv: torch.Tensor, n_heads: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``dtype`` None: the arithmetic follows the INPUT -- bf16 kernels for a bf16 projection (what ``nn.Linear`` emits
    under ``torch.autocast(bfloat16)``), exact-fp32 kernels for an fp32 one (the reference's un-autocast
    ``train_transformer`` loop, train.py:150).  Pass ``dtype`` to force one."""
    if dtype is None:
        dtype = qkv.dtype
    if dtype not in _DT:
        raise RuntimeError(f"causal_attention: dtype {dtype} not supported (float32 / bfloat16)")
    b, s, d3 = qkv.shape
    hd = d3 // 3 // n_heads
    if hd in _ATTN_HEAD_DIMS:
        return _CausalAttention.apply(qkv, n_heads, dtype)
    if hd > _ATTN_HEAD_DIMS[-1]:                                    # wider than any kernel instantiation: ATen's fused attention on the GPU
        q, k, v = (t.reshape(b, s, n_heads, hd).transpose(1, 2).to(dtype) for t in qkv.split(n_heads * hd, dim=-1))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
        return o.transpose(1, 2).reshape(b, s, n_heads * hd).to(qkv.dtype)
    # any other head width (the reference's constructor takes every hidden_dim divisible by the head count, models/transformer.py:17-35): the
    # heads are zero-padded to the next width the kernels have -- zero dimensions add nothing to a score, produce zero context columns and
    # receive zero gradients -- with the softmax scale of the TRUE width.  Differentiable torch ops around the same node; off the benched path.
    hp = next(v for v in _ATTN_HEAD_DIMS if v >= hd)
    x = torch.nn.functional.pad(qkv.reshape(b, s, 3, n_heads, hd), (0, hp - hd)).reshape(b, s, 3 * n_heads * hp)
    o = _CausalAttention.apply(x, n_heads, dtype, float(hd) ** -0.5)
    return o.reshape(b, s, n_heads, hp)[..., :hd].reshape(b, s, n_heads * hd)


def attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, past: int, n_heads: int) -> torch.Tensor:
    """Inference-only cached attention (``mas_attn_decode``): q [B,nq,H*hd] (the new positions), k_cache / v_cache [B,S_max,H*hd]
    whose rows 0 .. past+nq-1 are valid (the new keys / values already appended).  Query i attends to keys 0 .. past+i.
    Returns the context [B,nq,H*hd]."""
    _require_cuda(q, "attention_decode")
    if q.dtype not in _DT or k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise RuntimeError("attention_decode: q / k / v must share a dtype in {float32, bfloat16}")
    b, nq, d = q.shape
    hd = d // n_heads
    if k_cache.shape[0] != b or k_cache.shape[2] != d or v_cache.shape != k_cache.shape or past + nq > k_cache.shape[1]:
        raise RuntimeError(f"attention_decode: cache {tuple(k_cache.shape)} does not hold past={past} + nq={nq} rows of width {d}")
    if q.stride(2) != 1 or k_cache.stride(2) != 1 or v_cache.stride(2) != 1:
        raise RuntimeError("attention_decode: the last dimension must be contiguous")
    if hd not in _ATTN_HEAD_DIMS:
        # head widths the decode kernel does not have (see ``causal_attention``): the same arithmetic from ATen ops on the GPU (inference only)
        L = past + nq
        qh = q.reshape(b, nq, n_heads, hd).transpose(1, 2).float()
        kh = k_cache[:, :L].reshape(b, L, n_heads, hd).transpose(1, 2).float()
        vh = v_cache[:, :L].reshape(b, L, n_heads, hd).transpose(1, 2).float()
        sc = torch.matmul(qh, kh.transpose(-1, -2)) * (float(hd) ** -0.5)
        allowed = torch.arange(L, device=q.device)[None, :] <= (past + torch.arange(nq, device=q.device))[:, None]
        pr = torch.softmax(sc.masked_fill(~allowed, float("-inf")), dim=-1)
        return torch.matmul(pr, vh).transpose(1, 2).reshape(b, nq, d).to(q.dtype)
    o = torch.empty((b, nq, d), dtype=q.dtype, device=q.device)
    check(lib().mas_attn_decode(_ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(o), _DT[q.dtype], b, n_heads, nq, int(past), hd,
                                q.stride(1), k_cache.stride(1), v_cache.stride(1), o.stride(1), q.stride(0), k_cache.stride(0),
                                v_cache.stride(0), o.stride(0), float(hd) ** -0.5, _stream()), "attn_decode")
    return o


# --------------------------------------------------------------------------- #
# transformer row operators: tanh-GELU, LayerNorm (+ fused residual)
# --------------------------------------------------------------------------- #
def _check_dtype(t: torch.Tensor, what: str):
    if t.dtype not in _DT:
        raise RuntimeError(f"{what}: dtype {t.dtype} not supported (float32 / bfloat16)")


_GELU_COLSUM = os.environ.get("MAS_GELU_COLSUM", "1") == "1"


class _GeluTanh(torch.autograd.Function):
    """OpenAI tanh-GELU (reference models/transformer.py:11-14), one streaming pass each way.  When x is what a ``_LinearBf16`` node
    returned (``lin1`` of the MLP, :125,129) the backward pass also sums the dx it writes over the rows and leaves the result for that
    layer's bias gradient (``mas_gelu_tanh_bwd_colsum`` + ``_ColsumHint``, as the sandwich LayerNorms do for ``out_proj`` / ``lin2``)."""

    @staticmethod
    def forward(ctx, x, want_colsum=False):
        _require_cuda(x, "gelu_tanh")
        _check_dtype(x, "gelu_tanh")
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().mas_gelu_tanh_fwd(_ptr(x), _ptr(y), _DT[x.dtype], x.numel(), _stream()), "gelu_tanh_fwd")
        ctx.save_for_backward(x)
        ctx.want_colsum = bool(want_colsum)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        cols = x.shape[-1]
        rows = x.numel() // cols
        wsb = lib().mas_gelu_tanh_bwd_colsum_workspace(_DT[x.dtype], cols) if ctx.want_colsum else 0
        if wsb:
            dc = torch.empty(cols, dtype=torch.float32, device=x.device)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
            check(lib().mas_gelu_tanh_bwd_colsum(_ptr(x), _ptr(dy), _ptr(dx), _ptr(dc), _DT[x.dtype], rows, cols, _ptr(ws), wsb, _stream()),
                  "gelu_tanh_bwd_colsum")
            _colsum_hint.put(dx, dc)
        else:
            check(lib().mas_gelu_tanh_bwd(_ptr(x), _ptr(dy), _ptr(dx), _DT[x.dtype], x.numel(), _stream()), "gelu_tanh_bwd")
        return dx, None


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    want = _GELU_COLSUM and x.dim() >= 2 and x.grad_fn is not None and type(x.grad_fn).__name__.startswith("_LinearBf16")
    return _GeluTanh.apply(x, want)


class _LayerNorm(torch.autograd.Function):
    """y = [residual +] LayerNorm(x) over the last dimension (reference TransformerLayer, models/transformer.py:197-210).
    x / dx have the input's dtype, y / residual / dy have ``out_dtype``; statistics and arithmetic are fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, eps, out_dtype, want_colsum=False):
        _require_cuda(x, "layer_norm")
        _check_dtype(x, "layer_norm")
        d = x.shape[-1]
        rows = x.numel() // d
        x = x.contiguous()
        if residual is not None:
            if residual.shape != x.shape:
                raise RuntimeError("layer_norm: residual shape must equal the input shape")
            residual = residual.to(out_dtype).contiguous()
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mr = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        check(lib().mas_layernorm_fwd(_ptr(x), _ptr(w32), _ptr(b32), _ptr(residual), _ptr(y), _ptr(mr), _DT[x.dtype],
                                      _DT[out_dtype], rows, d, float(eps), _stream()), "layernorm_fwd")
        ctx.save_for_backward(x, weight, mr)
        ctx.has_res, ctx.out_dtype, ctx.want_colsum = residual is not None, out_dtype, bool(want_colsum)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mr = ctx.saved_tensors
        dx, dg, db = _layer_norm_bwd(x, weight, mr, dy.to(ctx.out_dtype).contiguous(), ctx.out_dtype, None, ctx.want_colsum)
        return dx, dg, db, (dy.to(ctx.out_dtype) if ctx.has_res else None), None, None, None


class _ColsumHint:
    """The column sums of ONE gradient tensor, computed by the kernel that wrote it, for the node that receives it next.

    A LayerNorm whose input is the output of a Linear layer (``LayerNorm.forward(..., producer_bias_grad=True)``: the sandwich
    LayerNorms behind ``out_proj`` / ``lin2``) accumulates the column sums of its dx inside its backward kernel
    (``mas_layernorm_bwd_colsum``); ``_LinearBf16.backward`` asks here before it launches ``mas_colsum`` on that same tensor.  One slot
    per (device, stream) -- the sums are produced on the LayerNorm's stream and may only be consumed on the same one: the entry holds a
    STRONG reference to dx, so the allocator cannot hand the same address to another tensor while the entry exists, and a hit requires
    the same storage address, element count, column count and version counter (no in-place write since) -- anything else (a Dropout in
    between, an accumulation, a hook that replaced the gradient) misses and the Linear computes its own sums.  Entries never outlive the
    backward pass that made them: ``put`` queues an end-of-backward callback on the autograd engine that empties the table (a pruned
    graph, ``autograd.grad`` on an intermediate or an exception would otherwise keep the last dx -- 25 MB at 12288 x 1024 -- alive)."""

    def __init__(self):
        self.slots = {}
        self.hits = 0                                # (tests / probes read this)
        self._cb_pending = False

    @staticmethod
    def _key(t):
        return (t.device.index, torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else (None, 0)

    def _end_of_backward(self):
        self.slots.clear()
        self._cb_pending = False

    def put(self, dx, sums):
        self.slots[self._key(dx)] = (dx, dx.data_ptr(), dx.numel(), dx.shape[-1], dx._version, sums)
        if not self._cb_pending:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self._cb_pending = True
            except RuntimeError:                     # not inside a backward pass (direct call of the raw function): the slot is taken or
                pass                                 # overwritten by the next put

    def take(self, dy2):
        ent = self.slots.pop(self._key(dy2), None)
        if ent is None:
            return None
        dx, ptr, numel, cols, version, sums = ent
        if (dy2.data_ptr() == ptr and dy2.numel() == numel and dy2.shape[-1] == cols and dy2.dtype == dx.dtype and dy2.device == dx.device
                and dy2._version == version and dx._version == version):
            self.hits += 1
            return sums
        return None

    def clear(self):
        self.slots.clear()


_colsum_hint = _ColsumHint()


def _layer_norm_bwd(x, weight, mr, dy, out_dtype, dx_add, want_colsum=False):
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dg = torch.empty(d, dtype=torch.float32, device=x.device)
    db = torch.empty(d, dtype=torch.float32, device=x.device)
    dc = torch.empty(d, dtype=torch.float32, device=x.device) if want_colsum else None
    wsb = lib().mas_layernorm_bwd_workspace(rows, d)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
    w32 = weight.detach().float().contiguous()
    check(lib().mas_layernorm_bwd_colsum(_ptr(x), _ptr(dy), _ptr(w32), _ptr(mr), _ptr(dx_add), _ptr(dx), _ptr(dg), _ptr(db), _ptr(dc),
                                         _DT[x.dtype], _DT[out_dtype], rows, d, _ptr(ws), wsb, _stream()), "layernorm_bwd")
    if want_colsum:
        _colsum_hint.put(dx, dc)
    return dx, dg.to(weight.dtype), db.to(weight.dtype)


class _LayerNormFork(torch.autograd.Function):
    """(LayerNorm(x), x): the pre-LayerNorm of a transformer block together with the skip connection that leaves the same tensor
    (reference models/transformer.py:197-210: ``x`` feeds ``ln_in`` AND the residual of the sandwich LayerNorm).  As one node the
    two gradients that come back to x are added inside the LayerNorm backward kernel (``mas_layernorm_bwd_add``) instead of by a
    separate elementwise pass of the autograd engine."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        _require_cuda(x, "layer_norm")
        _check_dtype(x, "layer_norm")
        d = x.shape[-1]
        rows = x.numel() // d
        x = x.contiguous()
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mr = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        check(lib().mas_layernorm_fwd(_ptr(x), _ptr(w32), _ptr(b32), None, _ptr(y), _ptr(mr), _DT[x.dtype], _DT[out_dtype], rows, d,
                                      float(eps), _stream()), "layernorm_fwd")
        ctx.save_for_backward(x, weight, mr)
        ctx.out_dtype = out_dtype
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, weight, mr = ctx.saved_tensors
        if dy is None:                                     # only the skip connection was used
            return dskip, None, None, None, None
        if dskip is not None:
            dskip = dskip.to(x.dtype).contiguous()
        dx, dg, db = _layer_norm_bwd(x, weight, mr, dy.to(ctx.out_dtype).contiguous(), ctx.out_dtype, dskip)
        return dx, dg, db, None, None


class _LayerNormPair(torch.autograd.Function):
    """(LN2(xnew), xnew) with xnew = residual + LN1(h): the sandwich LayerNorm + residual of one sub-block and the pre-LayerNorm of the
    next as ONE node (``mas_layernorm_pair_fwd``; reference models/transformer.py:201-203 + :205, and
    :207-209 + the next layer's :197 / the final LayerNorm).  ``xnew`` is also the skip connection of the next sub-block, so the node has
    the two outputs ``_LayerNormFork`` has.  Forward: one launch (``mas_layernorm_pair_fwd``: the row stays in registers between the two
    normalisations, 12 B per element instead of 16).  Backward: the two launches of the unfused form.  Values: bit for bit those of
    ``_LayerNorm`` followed by ``_LayerNormFork``."""

    @staticmethod
    def forward(ctx, h, residual, w1, b1, eps1, w2, b2, eps2, y_dtype, want_colsum):
        _require_cuda(h, "layer_norm_pair")
        d = h.shape[-1]
        rows = h.numel() // d
        h = h.contiguous()
        residual = residual.contiguous()
        xnew = torch.empty(h.shape, dtype=residual.dtype, device=h.device)
        y2 = torch.empty(h.shape, dtype=y_dtype, device=h.device)
        mr1 = torch.empty((rows, 2), dtype=torch.float32, device=h.device)
        mr2 = torch.empty((rows, 2), dtype=torch.float32, device=h.device)
        w1f, b1f = w1.detach().float().contiguous(), b1.detach().float().contiguous()
        w2f, b2f = w2.detach().float().contiguous(), b2.detach().float().contiguous()
        check(lib().mas_layernorm_pair_fwd(_ptr(h), _ptr(w1f), _ptr(b1f), _ptr(residual), _ptr(w2f), _ptr(b2f), _ptr(xnew), _ptr(y2), _ptr(mr1),
                                           _ptr(mr2), _DT[h.dtype], _DT[residual.dtype], _DT[y_dtype], rows, d, float(eps1), float(eps2), _stream()),
              "layernorm_pair_fwd")
        ctx.save_for_backward(h, xnew, w1, w2, mr1, mr2)
        ctx.y_dtype, ctx.want_colsum = y_dtype, bool(want_colsum)
        return y2, xnew

    @staticmethod
    def backward(ctx, dy2, dskip):
        h, xnew, w1, w2, mr1, mr2 = ctx.saved_tensors
        d = h.shape[-1]
        rows = h.numel() // d
        if dskip is not None:
            dskip = dskip.to(xnew.dtype).contiguous()
        if dy2 is None:                                    # the second LayerNorm's output was not used: LN1's backward alone
            if dskip is None:
                return (None,) * 10
            dh, dg1, db1 = _layer_norm_bwd(h, w1, mr1, dskip, xnew.dtype, None, ctx.want_colsum)
            return dh, dskip, dg1, db1, None, None, None, None, None, None
        # LN2's backward with the skip gradient added in the kernel -> the gradient of xnew (= of the residual), then LN1's backward on it
        # (with the producer's bias gradient): the two launches of the unfused form -- the fused backward kernel lost (transformer_ew.hip)
        dxn, dg2, db2 = _layer_norm_bwd(xnew, w2, mr2, dy2.to(ctx.y_dtype).contiguous(), ctx.y_dtype, dskip)
        dh, dg1, db1 = _layer_norm_bwd(h, w1, mr1, dxn, xnew.dtype, None, ctx.want_colsum)
        return dh, dxn, dg1, db1, None, dg2, db2, None, None, None


_LN_PAIR = os.environ.get("MAS_LN_PAIR", "1") == "1"


def layer_norm_pair(h, residual, ln1, ln2, producer_bias_grad=False):
    """-> (ln2(xnew), xnew) with xnew = residual + ln1(h), for two ``nn.LayerNorm``-like modules over the last dimension.  One fused
    forward launch where the kernel applies -- the autocast transformer (h bf16, fp32 residual stream, bf16 consumer) or everything fp32,
    D % 4 == 0, D <= 1024, equal shapes -- else (and with ``MAS_LN_PAIR=0``) the two separate nodes: the same values either way."""
    y_dtype = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else residual.dtype
    d = h.shape[-1]
    ok = (_LN_PAIR and h.is_cuda and residual.shape == h.shape and residual.dtype == torch.float32 and d % 4 == 0 and d <= 1024 and _ln_ok(d, h.dtype, residual.dtype)
          and ((h.dtype == torch.bfloat16 and y_dtype == torch.bfloat16) or (h.dtype == torch.float32 and y_dtype == torch.float32)))
    if not ok:
        xnew = layer_norm(h, ln1.weight, ln1.bias, ln1.eps, residual, producer_bias_grad=producer_bias_grad)
        return layer_norm_fork(xnew, ln2.weight, ln2.bias, ln2.eps)
    want = bool(producer_bias_grad) and h.grad_fn is not None and type(h.grad_fn).__name__.startswith("_LinearBf16")
    return _LayerNormPair.apply(h, residual, ln1.weight, ln1.bias, ln1.eps, ln2.weight, ln2.bias, ln2.eps, y_dtype, want)


def _ln_ok(d: int, in_dtype, out_dtype) -> bool:
    """the LayerNorm kernels' envelope (transformer_ew.hip ln_check): a whole number of 16-byte vectors per row, at most 256 of them; other
    widths (hidden_dim 100, say: the reference's constructor takes any) run ATen's layer_norm on the GPU -- off the benched path"""
    if in_dtype not in _DT or out_dtype not in _DT:
        return False
    vec = 8 if (in_dtype == torch.bfloat16 and out_dtype == torch.bfloat16) else 4
    return d % vec == 0 and d <= 64 * vec * 4


def layer_norm(x, weight, bias, eps=1e-5, residual=None, out_dtype=None, producer_bias_grad=False):
    """``out_dtype`` None: the residual's dtype if one is given, else the autocast dtype when autocast is on (the
    consumer is a Linear that would cast anyway), else the input's dtype.  ``producer_bias_grad``: x is the output of a Linear layer --
    the backward kernel also sums its dx over the rows and leaves the result for that layer's bias gradient (``_ColsumHint``)."""
    if out_dtype is None:
        if residual is not None:
            out_dtype = residual.dtype
        elif torch.is_autocast_enabled():
            out_dtype = torch.get_autocast_gpu_dtype()
        else:
            out_dtype = x.dtype
    if out_dtype not in _DT:
        raise RuntimeError(f"layer_norm: output dtype {out_dtype} not supported")
    if not _ln_ok(x.shape[-1], x.dtype, out_dtype):
        y = torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps)      # widths outside the kernels' envelope
        return (y + residual.float() if residual is not None else y).to(out_dtype)
    # (only when x really is what a ``_LinearBf16`` node returned: any other producer -- nn.Linear outside autocast, a Dropout with
    # p > 0, a prescale -- would never take the sums, and the slot would keep dx alive for nothing)
    want = bool(producer_bias_grad) and x.grad_fn is not None and type(x.grad_fn).__name__.startswith("_LinearBf16")
    return _LayerNorm.apply(x, weight, bias, residual, eps, out_dtype, want)


def layer_norm_fork(x, weight, bias, eps=1e-5, out_dtype=None):
    """-> (LayerNorm(x), x): use the second output wherever the block adds x back (see ``_LayerNormFork``)."""
    if out_dtype is None:
        out_dtype = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else x.dtype
    if out_dtype not in _DT:
        raise RuntimeError(f"layer_norm: output dtype {out_dtype} not supported")
    if not _ln_ok(x.shape[-1], x.dtype, out_dtype):
        return torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps).to(out_dtype), x
    return _LayerNormFork.apply(x, weight, bias, eps, out_dtype)


# --------------------------------------------------------------------------- #
# Linear layers of the transformer under bf16 autocast: library GEMMs + the HIP column-sum for the bias gradient
# --------------------------------------------------------------------------- #
def colsum(x2d: torch.Tensor) -> torch.Tensor:
    """[rows, cols] bf16 / fp32 (contiguous) -> fp32 [cols] column sums (``mas_colsum``: fixed summation order)."""
    _require_cuda(x2d, "colsum")
    if x2d.dim() != 2 or not x2d.is_contiguous() or x2d.dtype not in _DT:
        raise RuntimeError("colsum: a contiguous 2-D bf16 / fp32 matrix")
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    wsb = lib().mas_colsum_workspace(rows, cols)
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=x2d.device)
    check(lib().mas_colsum(_ptr(x2d), _DT[x2d.dtype], rows, cols, _ptr(out), _ptr(ws), wsb, _stream()), "colsum")
    return out


class _Bf16Shadows:
    """bf16 copies of the fp32 parameters of every registered Linear layer, refreshed TOGETHER: the first layer that finds its copy
    stale (an optimizer step or an in-place write changed its stamp) recasts every stale parameter in one multi-tensor launch
    (``torch._foreach_copy_``) instead of ~200 small cast kernels per step; between optimizer steps (gradient accumulation over
    micro-batches) nothing is recast at all.  Same validity rule as the packed conv weights: writes through ``.data`` need
    ``invalidate_weight_cache()``."""

    def __init__(self):
        self.params = {}                             # id -> (weakref, [version, data_ptr], bf16 copy)

    def clear(self):
        self.params.clear()

    def drop(self, p):
        ent = self.params.get(id(p))
        if ent is not None:
            ent[1] = None

    def register(self, p: torch.nn.Parameter):
        k = id(p)
        if k not in self.params:
            self.params[k] = [weakref.ref(p, lambda _r, k=k: self.params.pop(k, None)), None, None]

    def get(self, p: torch.nn.Parameter) -> torch.Tensor:
        self.register(p)
        ent = self.params[id(p)]
        if ent[1] == _param_stamp(p) and ent[2] is not None and ent[2].device == p.device:
            return ent[2]
        srcs, dsts = [], []
        for e in list(self.params.values()):
            q = e[0]()
            if q is None or not q.is_cuda or q.device != p.device or q.dtype != torch.float32:
                continue
            if e[1] != _param_stamp(q) or e[2] is None or e[2].device != q.device:
                if e[2] is None or e[2].shape != q.shape or e[2].device != q.device:
                    e[2] = torch.empty(q.shape, dtype=torch.bfloat16, device=q.device)
                e[1] = _param_stamp(q)
                srcs.append(q.detach()); dsts.append(e[2])
        if dsts:
            torch._foreach_copy_(dsts, srcs)
        return ent[2]


_bf16_shadows = _Bf16Shadows()


class _LinearBf16(torch.autograd.Function):
    """y = x W^T + b with bf16 operands / fp32 accumulation (what ``torch.autocast(bfloat16)`` makes of nn.Linear, reference
    models/transformer.py:31,34,125,126), as ONE autograd node: forward and the two backward products are the library GEMMs,
    the bias gradient is ``mas_colsum`` (fp32, fixed order) instead of a generic reduction + cast; the weight gradient is the
    GEMM's bf16 result cast to fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        with torch.autocast("cuda", enabled=False):
            k = x.shape[-1]
            x2 = x.reshape(-1, k)
            x2 = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
            if isinstance(weight, torch.nn.Parameter) and isinstance(bias, torch.nn.Parameter) and weight.dtype == torch.float32:
                wb, bb = _bf16_shadows.get(weight), _bf16_shadows.get(bias)
            else:
                wb, bb = weight.detach().to(torch.bfloat16), bias.detach().to(torch.bfloat16)
            y = torch.addmm(bb, x2, wb.t())
        ctx.save_for_backward(x2, wb)
        ctx.in_shape, ctx.in_dtype = x.shape, x.dtype
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wb = ctx.saved_tensors
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, dy.shape[-1])
            dy2 = (dy2 if dy2.dtype == torch.bfloat16 else dy2.to(torch.bfloat16)).contiguous()
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = torch.mm(dy2, wb).view(ctx.in_shape).to(ctx.in_dtype)
            if ctx.needs_input_grad[1]:
                dw = torch.mm(dy2.t(), x2).float()     # (fp32 straight from the accumulators is outside TunableOp: +1 ms per step, DESIGN history R2)
            if ctx.needs_input_grad[2]:
                db = _colsum_hint.take(dy2)             # (the LayerNorm backward that wrote this very tensor summed it)
                if db is None:
                    db = colsum(dy2)
            else:
                _colsum_hint.clear()
        return dx, dw, db


def linear_bf16(x, weight, bias):
    return _LinearBf16.apply(x, weight, bias)
