"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mas_hip.h declares; the Python surface mirrors the reference's class surface (import paths,
constructor kwargs, state_dict keys); the product path refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mas_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mas_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import mas_hip
    if not os.path.exists(mas_hip.LIB_PATH):
        from mas_hip import build
        build.build(verbose=False)
    L = ctypes.CDLL(mas_hip.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mas_hip.h but not exported"
    assert sorted(mas_hip.EXPORTS) == syms, "ctypes binding and header disagree"
    assert mas_hip.lib().mas_abi_version() == mas_hip.ABI_VERSION


def test_argument_validation_without_gpu():
    """error convention: negative code + message, no exception across the ABI, no compute without a GPU"""
    import mas_hip
    L = mas_hip.lib()
    assert L.mas_conv_fwd(None, None, None, None, None, None, None, None) == -1
    assert b"null" in L.mas_last_error()
    with pytest.raises(RuntimeError):
        mas_hip.check(-1, "probe")
    assert L.mas_packed_weight_elems(128, 128, 3) == 9 * 128 * 128
    assert L.mas_packed_weight_elems(3, 128, 3) == 9 * 128 * 128    # Cout padded to the 128-row tile


def test_round2_entry_points_validate_arguments_without_gpu():
    """the ABI v2 additions follow the same convention: negative code + message, nothing computed"""
    import mas_hip
    L = mas_hip.lib()
    assert L.mas_abi_version() == mas_hip.ABI_VERSION == 9
    assert L.mas_conv_weight_layout(None) == mas_hip.WLAYOUT_K64
    d = mas_hip.ConvDesc(32, 256, 256, 128, 256, 256, 128, 3, 1, 1, 1, mas_hip.BF16, mas_hip.BF16, 0, 0, 0)
    assert L.mas_conv_weight_layout(ctypes.byref(d)) in (mas_hip.WLAYOUT_K64, mas_hip.WLAYOUT_K32)
    assert L.mas_conv_stat_rows(ctypes.byref(d)) == 0                       # a K64 image never takes the fused-statistics kernel
    d.w_layout = 7
    assert L.mas_conv_fwd(ctypes.byref(d), 1, None, 1, None, None, 1, None) == -1 and b"w_layout" in L.mas_last_error()
    assert L.mas_attn_decode(None, None, None, None, 1, 1, 1, 1, 0, 64, 64, 64, 64, 64, 0, 0, 0, 0, 0.125, None) == -1
    assert L.mas_spatial_attn_fwd(1, 1, None, mas_hip.F32, 1, 16, 64, None) == -2 and b"bf16" in L.mas_last_error()
    assert L.mas_spatial_attn_fwd(1, 1, None, mas_hip.BF16, 1, 300, 64, None) == -2    # more than 256 tokens
    assert L.mas_pack_conv_weight_layout(1, 1, 128, 128, 3, 0, mas_hip.F32, mas_hip.WLAYOUT_K32, None) == -2   # K32 is bf16 only
    assert L.mas_space_to_depth2x(None, None, 1, 1, 2, 2, 8, 2, 2, 1, None) == -1
    # ABI v6: Upsample + conv in its sub-pixel form (conv_up2.hip)
    assert L.mas_packed_weight_elems_up2(128, 128) == 16 * 128 * 128 and L.mas_packed_weight_elems_up2(128, 512) == 16 * 128 * 512
    assert L.mas_conv_up2_supported(None) == 0 and L.mas_conv_up2_dgrad_supported(None) == 0
    assert L.mas_conv_up2_dgrad(None, None, None, None, None) == -1
    assert L.mas_pack_conv_weight_layout(1, 1, 128, 128, 3, 0, mas_hip.F32, mas_hip.WLAYOUT_UP2, None) == -2   # UP2 is bf16 / 3x3 only
    assert L.mas_pack_batch_blocks(128, 128, 3, 0, mas_hip.BF16, mas_hip.WLAYOUT_UP2) == 128      # 16 tap tiles x 4 chunks x 128 rows x 4 slots / 256
    plain = mas_hip.ConvDesc(32, 128, 128, 128, 128, 128, 128, 3, 1, 1, 1, mas_hip.BF16, mas_hip.BF16, 0, 0, 0)
    assert L.mas_conv_up2_supported(ctypes.byref(plain)) == 0               # not an Upsample convolution


def test_surface_matches_reference_contract():
    from models import VQBASE
    from models.modules import Encoder, Decoder, Codebook, ResnetBlock, AttnBlock, Upsample, Downsample  # noqa: F401
    cfg = dict(ddconfig=dict(z_channels=256, in_channels=3, out_channels=3, channels=[128, 128, 128, 256, 512, 512],
                             num_res_blocks=2, resolution=512, attn_resolutions=[32], dropout=0.0),
               n_embed=8192, embed_dim=256, init_steps=3000, reservoir_size=12500)   # conf/img_config.yaml:19-34
    m = VQBASE(**cfg)
    sd = m.state_dict()
    assert len(sd) == 348 and sum(p.numel() for p in m.parameters()) == 95219075
    from oracle.vq_oracle import synth_state_dict
    ref_keys = synth_state_dict(cfg["ddconfig"], 8192, 256)                         # strict-loaded into the reference in make_golden.py
    assert set(sd) == set(ref_keys) and all(sd[k].shape == ref_keys[k].shape for k in sd)
    assert hasattr(m.decoder.model[-1], "weight") and m.quantize.q_counter == 0
    assert all(type(p) is torch.nn.Parameter for p in m.parameters())
    # the loss stack and stage-2 helpers resolve under the reference's import paths too
    import losses
    from losses.loss_img import VQLPIPSWithDiscriminator, hinge_d_loss, vanilla_d_loss, adopt_weight  # noqa: F401
    from losses.discriminator import Discriminator, weights_init  # noqa: F401
    assert losses.VQLPIPSWithDiscriminator is VQLPIPSWithDiscriminator
    assert len(Discriminator().state_dict()) == 22
    assert hasattr(m, "encode_to_indices") and hasattr(m, "decode_code")


def test_no_cpu_fallback():
    from models import VQBASE
    cfg = dict(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64], num_res_blocks=1,
                             resolution=16, attn_resolutions=[8], dropout=0.0), n_embed=64, embed_dim=32, init_steps=10, reservoir_size=100)
    m = VQBASE(**cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 16, 16))


def test_optimizer_steps_are_part_of_the_parameter_stamp():
    """fused optimizers do not bump Parameter._version; the stamp that validates packed weights / bf16 shadows also counts optimizer
    steps through a process-wide post-step hook (CPU-checkable half of tests/test_gpu_parity_r2.py's fused-Adam test)."""
    import torch
    from mas_hip import ops
    p, q = torch.nn.Parameter(torch.randn(3, 3)), torch.nn.Parameter(torch.randn(3))
    p.grad, q.grad = torch.randn(3, 3), torch.randn(3)
    opt = torch.optim.SGD([p], lr=0.1)
    sp, sq = ops._param_stamp(p), ops._param_stamp(q)
    opt.step()
    assert ops._param_stamp(p)[2] == sp[2] + 1 and ops._param_stamp(q) == sq        # only the optimizer's own parameters are marked


def test_losses_package_needs_no_reference_checkout():
    """Round 6 (VERDICT r5 missing #4): ``losses.loss_seg`` is this package's own module -- both import forms a ``_target_:`` string
    can take resolve in a fresh interpreter whose only path entry is the package, and nothing in the package looks for a checkout."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "make-a-scene_amd")
    code = ("import sys; sys.path.insert(0, %r); from losses.loss_seg import VQVAEWithBCELoss as A; import losses.loss_seg as M; import losses; "
            "assert M.__file__.startswith(%r) and losses.VQVAEWithBCELoss is A and losses.BCELossWithQuant is M.BCELossWithQuant; "
            "print('own loss_seg ok')" % (pkg, pkg))
    env = {k: v for k, v in os.environ.items() if k != "MAS_REFERENCE_ROOT"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/")
    assert r.returncode == 0 and "own loss_seg ok" in r.stdout, r.stdout + r.stderr
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "MAS_REFERENCE_ROOT" not in open(os.path.join(dirpath, f)).read(), os.path.join(dirpath, f)
