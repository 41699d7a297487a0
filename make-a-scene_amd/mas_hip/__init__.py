"""ctypes binding of libmas_hip.so (the C ABI declared in include/mas_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel call
fails, a RuntimeError is raised.  PyTorch is used only for device memory, streams and
torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# HIP gives a process four hardware queues by default and hands them to streams in creation order; RCCL's own streams take the three beside
# the default stream's, after which ops' side stream (weight gradients beside the GroupNorm backward) lands on the default stream's queue and
# its kernels serialise behind barrier packets: +1.7 ms per step instead of -1.2 (profiles/r06_wgrad_stream.txt).  The runtime reads this at
# its first HIP call -- after `import torch` is early enough, after torch.cuda.is_available() is not (ops probes the streams and refuses the
# side stream when they do not overlap).  A value the user exported wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: F401,E402  -- must come first: PyTorch-ROCm bundles its own libamdhip64; loading ours before it leaves torch without GPUs

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAS_HIP_LIB") or os.path.join(_HERE, "libmas_hip.so")   # env override: A/B kernel experiments

F32, BF16 = 0, 1
ACT_NONE, ACT_AFFINE, ACT_AFFINE_SILU = 0, 1, 2
ABI_VERSION = 9
WLAYOUT_K64, WLAYOUT_K32, WLAYOUT_UP2 = 0, 1, 2


class ConvDesc(C.Structure):
    """Mirror of ``MasConvDesc`` (include/mas_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "ks", "stride", "pad_top", "pad_left",
        "in_dtype", "out_dtype", "act", "upsample", "w_layout")]


class PackItem(C.Structure):
    """Mirror of ``MasPackItem`` (include/mas_hip.h): one entry of the batched weight-pack table."""
    _fields_ = [("w_oihw", C.c_void_p), ("packed", C.c_void_p)] + [(n, C.c_int32) for n in (
        "Cout", "Cin", "ks", "transpose", "dtype", "layout", "first_block", "n_blocks")]


class PackTileItem(C.Structure):
    """Mirror of ``MasPackTileItem`` (include/mas_hip.h): one parameter and up to four bf16 images of it."""
    _fields_ = [("w_oihw", C.c_void_p), ("img", C.c_void_p * 4), ("transpose", C.c_int32 * 4), ("layout", C.c_int32 * 4)] + \
               [(n, C.c_int32) for n in ("n_img", "Cout", "Cin", "ks", "first_block", "pad_")]


class AdamItem(C.Structure):
    """Mirror of ``MasAdamItem`` (include/mas_hip.h): one fp32 parameter with its gradient and the two Adam moments."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_longlong), ("first_block", C.c_int32),
                ("pad_", C.c_int32)]


_p, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGNATURES = {
    "mas_abi_version": (C.c_int, []),
    "mas_last_error": (C.c_char_p, []),
    "mas_last_kernel": (C.c_char_p, []),
    "mas_packed_weight_elems": (_sz, [_i, _i, _i]),
    "mas_packed_weight_elems_up2": (_sz, [_i, _i]),
    "mas_pack_conv_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "mas_conv_weight_layout": (_i, [C.POINTER(ConvDesc)]),
    "mas_pack_conv_weight_layout": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mas_pack_batch_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "mas_adam_blocks": (_i, [C.c_longlong]),
    "mas_adam_multi": (_i, [_p, _i, _i, _f, _f, _f, _f, _f, C.c_double, C.c_double, _p]),
    "mas_pack_conv_weight_batch": (_i, [_p, _i, _i, _p]),
    "mas_pack_tile_blocks": (_i, [_i, _i, _i]),
    "mas_pack_conv_weight_tiles": (_i, [_p, _i, _i, _i, _p]),
    "mas_gn_stats_workspace": (_sz, [_i, _i]),
    "mas_gn_stats": (_i, [_p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "mas_gn_bwd_workspace": (_sz, [_i, _i]),
    "mas_gn_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "mas_gn_bwd_3pass": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "mas_gn_act": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "mas_gn_small_supported": (_i, [_i, _i, _i, _i]),
    "mas_gn_stats_act": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _i, _p, _p, _p]),
    "mas_conv_fwd": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p]),
    "mas_conv_stat_rows": (_i, [C.POINTER(ConvDesc)]),
    "mas_conv_fwd_stats": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    "mas_gn_stats_from_partials": (_i, [_p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "mas_conv_wgrad": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p]),
    "mas_wgrad_commit": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "mas_conv_s2_dgrad_supported": (_i, [C.POINTER(ConvDesc)]),
    "mas_conv_s2_dgrad": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p]),
    "mas_conv_up2_supported": (_i, [C.POINTER(ConvDesc)]),
    "mas_conv_up2_dgrad_supported": (_i, [C.POINTER(ConvDesc)]),
    "mas_conv_up2_dgrad": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p]),
    "mas_conv_up2_wgrad_splits": (_i, [C.POINTER(ConvDesc)]),
    "mas_conv_up2_wgrad_partial": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p]),
    "mas_wgrad_reduce_up2": (_i, [_p, _p, _i, _p, _p, _i, _i, _p]),
    "mas_conv_wgrad_splits": (_i, [C.POINTER(ConvDesc)]),
    "mas_conv_wgrad_partial": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p]),
    "mas_wgrad_reduce": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _p]),
    "mas_vq_workspace": (_sz, [_i, _i]),
    "mas_vq_argmin_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "mas_vq_bwd": (_i, [_p, _p, _p, _p, _p, _f, _i, _i, _i, _p, _p, _p]),
    "mas_attn_causal_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, C.c_longlong, C.c_longlong, C.c_longlong, _f, _p]),
    "mas_attn_causal_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "mas_spatial_attn_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "mas_spatial_attn_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mas_attn_decode": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _f, _p]),
    "mas_upsample2x": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "mas_sumpool2x": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "mas_zero_stuff2x": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mas_space_to_depth2x": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mas_gelu_tanh_fwd": (_i, [_p, _p, _i, C.c_longlong, _p]),
    "mas_gelu_tanh_bwd": (_i, [_p, _p, _p, _i, C.c_longlong, _p]),
    "mas_gelu_tanh_bwd_colsum_workspace": (_sz, [_i, _i]),
    "mas_gelu_tanh_bwd_colsum": (_i, [_p, _p, _p, _p, _i, C.c_longlong, _i, _p, _sz, _p]),
    "mas_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "mas_layernorm_bwd_workspace": (_sz, [_i, _i]),
    "mas_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "mas_layernorm_bwd_add": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "mas_bn_workspace": (_sz, [_i, _i]),
    "mas_bn_partial_sums": (_i, [_p, _p, _p, _i, _i, _p, _p, _sz, _p]),
    "mas_bn_finalize": (_i, [_p, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p]),
    "mas_bn_apply": (_i, [_p, _p, _p, _i, _i, _p]),
    "mas_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "mas_bn_partial_sums_act": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _p, _p, _sz, _p]),
    "mas_bn_apply_act": (_i, [_p, _p, _p, _f, _i, _i, _i, _p]),
    "mas_bn_bwd_apply_act": (_i, [_p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _p]),
    "mas_layernorm_bwd_colsum": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "mas_layernorm_pair_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p]),
    "mas_colsum_workspace": (_sz, [_i, _i]),
    "mas_colsum": (_i, [_p, _i, _i, _i, _p, _p, _sz, _p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


def lib():
    """Loads libmas_hip.so once; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                        "(hipcc --offload-arch=gfx950); there is no PyTorch/CPU fallback for this path")
                L = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(L, name, None)
                    if fn is None:
                        continue          # optional symbol of a later ABI revision; callers check
                    fn.restype, fn.argtypes = res, args
                if L.mas_abi_version() != ABI_VERSION:
                    raise RuntimeError(f"libmas_hip.so ABI {L.mas_abi_version()} != binding {ABI_VERSION}")
                _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mas_last_error()
        raise RuntimeError(f"libmas_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")
