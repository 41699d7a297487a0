"""Round-3 parity holes (VERDICT r2 "What's weak" 1-3): the kernels bench.py times, ON the code path it times, against the oracle.

* the dominant convolution through the DEFAULT dispatch (unpacked weight -> K32 image -> conv3x3_wide.hip) at 128->128 @256x256
  with N=16: 2048 tiles on the 1024-work-group grid, every work-group walks two tiles (asserted) -- forward plain / GN+SiLU /
  GN+SiLU+residual, the data-gradient packing, and the LDS-DMA weight gradient, against F.conv2d / autograd in fp32 on the CPU;
* VQBASE (conf/img_config.yaml block) in bf16 at B=16 -- the smallest batch at which the 256x256 layers take the multi-tile walk --
  against oracle/vq_oracle.py on the same batch (latents, index agreement, decoder fed the oracle's z_q);
* the encoder BACKWARD with the reference's own dL/dz injected at quant_conv's output (tests/golden/vq_img256_bwd.npz), in fp32,
  in bf16 at B=1 and with the image replicated 16x (the benched wide-dgrad / wgrad-dma paths): kernel error without the index
  flips of bf16 latents, measured against the reference's OWN bf16-autocast deviation;
* MakeAScene at config 4's width (2 layers, d=1024, 16 heads, S=1536) against the reference's fp32 output
  (tests/golden/transformer_w1024.npz), in fp32 and under bf16 autocast.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

IMG = dict(ddconfig=dict(z_channels=256, in_channels=3, out_channels=3, channels=[128, 128, 128, 256, 512, 512],
                         num_res_blocks=2, resolution=512, attn_resolutions=[32], dropout=0.0),
           n_embed=8192, embed_dim=256, init_steps=3000, reservoir_size=12500)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relerr(got, ref):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def rel_l2(got, ref):
    got = got.detach().double().cpu()
    ref = torch.as_tensor(ref).double()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def _silu(u):
    return u * torch.sigmoid(u)


def _build(cfg, seed, dtype, train=True):
    from models import VQBASE
    from mas_hip import ops
    from oracle.vq_oracle import synth_state_dict
    ops.set_compute_dtype(dtype)
    m = VQBASE(**cfg)
    m.load_state_dict(synth_state_dict(cfg["ddconfig"], cfg["n_embed"], cfg["embed_dim"], seed=seed), strict=True)
    m = m.to(_dev()).train(train)
    m.quantize.q_counter = m.quantize.q_re_end
    return m


def _wide_grid_and_tiles(n, ho, wo, cout):
    """launch geometry of conv3x3_wide.hip (launch_wide): tiles of 16x32 pixels x 128 couts on min(tiles, 4 * CUs) work-groups"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = n * ((ho + 15) // 16) * ((wo + 31) // 32) * (cout // 128)
    return min(tiles, 4 * cus), tiles


# --------------------------------------------------------------------------------------------------------------
# 1. the dominant launch through the default dispatch, multi-tile walk
# --------------------------------------------------------------------------------------------------------------
def test_dominant_conv_wide_real_shape_vs_cpu_fp32(monkeypatch):
    """128->128 3x3 at 256x256, bf16, N=16, UNPACKED weight (ops.ConvWeight): the library picks the K32 image, i.e.
    conv3x3_wide_kernel, with 2048 tiles on 1024 work-groups.  Tolerances: bf16 outputs 1e-2 of max|ref| (0.4 % rounding of the
    largest value + fp32 accumulation order); fp32 weight / bias gradients 2e-3."""
    import mas_hip
    from mas_hip import ops
    dev = _dev()
    assert "MAS_CONV_WGS_PER_CU" not in os.environ and "MAS_CONV_WIDE" not in os.environ      # the default dispatch, as bench.py runs it
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    n, c, h = 16, 128, 256
    bf = torch.bfloat16
    d = ops._desc(n, h, h, c, h, h, c, 3, 1, 1, 1, bf, bf, 0, False)
    assert ops._preferred_layout(d) == mas_hip.WLAYOUT_K32
    grid, tiles = _wide_grid_and_tiles(n, h, h, c)
    assert tiles >= 2 * grid, (tiles, grid)                                                  # every work-group walks >= 2 tiles
    g = torch.Generator(device="cpu").manual_seed(33)
    x = torch.randn(n, c, h, h, generator=g).bfloat16()
    w = (torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)).bfloat16().float()
    b = 0.1 * torch.randn(c, generator=g)
    res = torch.randn(n, c, h, h, generator=g).bfloat16()
    dy = torch.randn(n, c, h, h, generator=g).bfloat16()
    ss = torch.stack([1.0 + 0.2 * torch.randn(n, c, generator=g), 0.3 * torch.randn(n, c, generator=g)], dim=-1).contiguous()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    xd, resd, dyd, ssd, wd, bd = cl(x), cl(res), cl(dy), ss.to(dev), w.to(dev), b.to(dev)
    geo = (h, h, c, h, h, c, 3, 1, 1, 1)
    xf = x.float()
    af = _silu(xf * ss[..., 0][:, :, None, None] + ss[..., 1][:, :, None, None]).bfloat16().float()

    seen = []
    ops.set_launch_hook(lambda kind, shape, launch: (seen.append((kind, shape)), launch()))
    try:
        y0 = ops.conv_fwd_raw(xd, None, ops.ConvWeight(wd, False), bd, None, n, *geo, 0, False, bf)
        y1 = ops.conv_fwd_raw(xd, ssd, ops.ConvWeight(wd, False), bd, None, n, *geo, 2, False, bf)
        y2 = ops.conv_fwd_raw(xd, ssd, ops.ConvWeight(wd, False), bd, resd, n, *geo, 2, False, bf)
        # data gradient: the parameter whose transpose=1 packing is `w` itself is P[o][i][kh][kw] = w[i][o][2-kh][2-kw]
        da = ops.conv_fwd_raw(dyd, None, ops.ConvWeight(w.permute(1, 0, 2, 3).flip(2, 3).contiguous().to(dev), True), None, None,
                              n, *geo, 0, False, bf)
        dw0, db0 = ops.conv_wgrad_raw(xd, None, dyd, n, *geo, 0, False, True)
        dw2, db2 = ops.conv_wgrad_raw(xd, ssd, dyd, n, *geo, 2, False, True)
        # the training path: the activation as a tensor (mas_gn_act); the convolution and the weight gradient run prologue-free on it
        a_out = ops.gn_act(xd, ssd, 2)
        y1a = ops.conv_fwd_raw(a_out, None, ops.ConvWeight(wd, False), bd, None, n, *geo, 0, False, bf)
        dw3, db3 = ops.conv_wgrad_raw(a_out, None, dyd, n, *geo, 0, False, True)
    finally:
        ops.set_launch_hook(None)
    torch.cuda.synchronize()
    assert [k for k, _ in seen] == ["conv_fwd"] * 4 + ["conv_wgrad"] * 2 + ["conv_fwd", "conv_wgrad"]
    e = relerr(a_out, af); print("materialised activation vs CPU silu(gn(x)) in bf16: %.3e" % e); assert e < 1e-2
    assert torch.equal(y1a, y1)                      # conv(gn_act(x)) == conv with the fused loader, bit for bit
    ref0 = F.conv2d(xf, w, b, padding=1)
    e = relerr(y0, ref0); print("wide fwd plain, 2 tiles per work-group: %.3e" % e); assert e < 1e-2
    del ref0
    ref1 = F.conv2d(af, w, b, padding=1)
    e = relerr(y1, ref1); print("wide fwd GN+SiLU: %.3e" % e); assert e < 1e-2
    e = relerr(y2, ref1 + res.float()); print("wide fwd GN+SiLU + residual: %.3e" % e); assert e < 1e-2
    del ref1
    refd = F.conv2d(dy.float(), w, None, padding=1)                        # `da` convolves dy with the EFFECTIVE filter w
    e = relerr(da, refd); print("wide dgrad packing: %.3e" % e); assert e < 1e-2
    del refd
    for act, a_in, dw, db in ((0, xf, dw0, db0), (2, af, dw2, db2), (3, af, dw3, db3)):      # 3: prologue-free on the materialised activation
        wr = torch.zeros(c, c, 3, 3, requires_grad=True)
        F.conv2d(a_in, wr, None, padding=1).backward(dy.float())
        e_w, e_b = relerr(dw, wr.grad), relerr(db, dy.float().sum((0, 2, 3)))
        print("wgrad (LDS-DMA kernel) act=%d N=16: dw %.3e db %.3e" % (act, e_w, e_b))
        assert e_w < 2e-3 and e_b < 2e-3


# --------------------------------------------------------------------------------------------------------------
# 2. the model at a batch where the benched kernels walk several tiles, vs the oracle on the same batch
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nb", [16, 32])
def test_img256_bf16_multi_tile_batch_vs_oracle(nb):
    """reference models/vqvae.py:36-39 through oracle/vq_oracle.py (pinned to the reference by tests/test_oracle_golden.py) on a
    16-image batch and (round 6, VERDICT r5 next #3b) on the 32-image batch bench.py runs; ours in the production precision.  Same split as the B=1 golden test: latents, index agreement, decoder fed
    the oracle's z_q.  The launch hook proves the 256x256 / 128x128 layers ran with more tiles than work-groups."""
    from mas_hip import ops
    from oracle import vq_oracle as O
    dev = _dev()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    x = O.synth_image_batch(nb, 3, 256, seed=6)
    sd = O.synth_state_dict(IMG["ddconfig"], IMG["n_embed"], IMG["embed_dim"], seed=1)
    taps = {}
    with torch.no_grad():
        ref, ref_q, ref_idx, ref_z = O.vqbase_forward(sd, x, IMG["ddconfig"], training=True, taps=taps)
    m = _build(IMG, 1, torch.bfloat16)
    got = {}
    m.quant_conv.register_forward_hook(lambda mod, i, o: got.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: got.__setitem__("q", o))
    shapes = []
    ops.set_launch_hook(lambda kind, shape, launch: (shapes.append((kind, shape)), launch()))
    try:
        with torch.no_grad():
            rec, q = m(x.to(dev))
            rec_ref_zq = m.decode(taps["z_q"].to(dev))
    finally:
        ops.set_launch_hook(None)
    torch.cuda.synchronize()
    walked = 0
    for kind, (n, h, w, cin, ho, wo, cout, ks, stride, act, has_res) in shapes:
        if kind == "conv_fwd" and ks == 3 and stride == 1 and cin % 64 == 0 and cout % 128 == 0 and wo >= 256:
            grid, tiles = _wide_grid_and_tiles(n, ho, wo, cout)
            assert tiles >= 2 * grid
            walked += 1
    assert walked >= 4 + 7 + 7                         # the 128->128 @256x256 layers: encoder once, decoder twice
    e_z, l2_z = relerr(got["z"], ref_z), rel_l2(got["z"], ref_z)
    agree = float((got["q"][2].cpu() == ref_idx).float().mean())
    e_dec, l2_dec = relerr(rec_ref_zq, ref), rel_l2(rec_ref_zq, ref)
    print(f"img256 bf16 B={nb} vs oracle: " "z max-rel %.3e rel-L2 %.3e | index agreement %.4f | decoder(oracle z_q) max-rel %.3e rel-L2 %.3e"
          " | q_loss %.5f vs %.5f" % (e_z, l2_z, agree, e_dec, l2_dec, float(q), float(ref_q)))
    assert e_z < 5e-2 and l2_z < 4e-2                  # the B=1 golden test's tolerances (52 bf16-storage layers)
    assert agree >= 0.93                               # measured 0.952 at B=16 (round 3); tightened from 0.90 in round 6
    assert e_dec < 5e-2 and l2_dec < 4e-2
    assert torch.isfinite(rec).all()


# --------------------------------------------------------------------------------------------------------------
# 3. encoder backward under the REFERENCE's dL/dz
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,copies", [("fp32", 1), ("bf16", 1), ("bf16", 16)])
def test_img256_encoder_backward_with_reference_dz(golden_dir, mode, copies):
    """dL/dz recorded from the reference's own backward (vq_img256_bwd.npz) is injected at quant_conv's output of OUR
    encoder: what differs from the reference's encoder gradients is then kernel / storage error only -- not the different codes
    a 1 % latent perturbation selects downstream (the 45 % seen on encoder.model.0.weight end to end in round 2).
    copies=16: the same image 16 times (BatchNorm's batch statistics are those of one copy; every parameter gradient is 16x the
    reference's), which moves the 256x256 / 128x128 layers onto the kernels and the multi-tile walk bench.py times.
    fp32 (exact-fp32 MFMA kernels): every recorded gradient within 2e-4 rel-L2 / 5e-4 max-rel (measured 1.7e-5 / 1.6e-5) and the
    gradient norm to 6 digits -- the kernels are right.
    bf16: GroupNorm's backward subtracts the group means, so most of each gradient cancels and rounding noise is amplified
    layer by layer towards the input (measured rel-L2: 3.0e-2 at quant_conv.0 ... 1.0e-1 at encoder.model.0.weight, 1.3e-1 at its
    bias).  The yardstick is the reference ITSELF under torch.autocast(bfloat16) on the CPU, recorded in the fixture
    (3.8e-2 ... 1.0e-1, 1.2e-1): each of our gradients must be within 1.2x (1.5x until round 5) of the reference's own bf16 deviation in rel-L2 and 2x in
    max-rel (the maximum over a 128-element GroupNorm weight gradient is a noisy statistic: a different summation order in ONE kernel
    moved encoder.model.1.norm1.weight from 0.8x to 1.57x of the reference's figure while its rel-L2 stayed at 1.006x; floor 5e-2),
    and the encoder gradient norm within 2 % of the fp32 reference."""
    sys.path.insert(0, golden_dir)
    from r3_spec import ENC_GRADS
    from oracle.vq_oracle import synth_image_batch
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "vq_img256_bwd.npz"))
    m = _build(IMG, 1, torch.float32 if mode == "fp32" else torch.bfloat16)
    x = synth_image_batch(1, 3, 256, seed=1).repeat(copies, 1, 1, 1).to(dev)
    z = m.quant_conv(m.encoder(x))
    z.backward(torch.from_numpy(g["dz"]).repeat(copies, 1, 1, 1).to(dev))
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    bad = []
    for k, sl in ENC_GRADS.items():
        got = params[k].grad.detach().float().cpu()[sl] / copies
        e2, em = rel_l2(got, g["grad:" + k]), relerr(got, g["grad:" + k])
        r2, rm = float(g["refbf16_l2:" + k]), float(g["refbf16_max:" + k])
        lim2, limm = (2e-4, 5e-4) if mode == "fp32" else (max(1.2 * r2, 5e-2), max(2.0 * rm, 5e-2))
        print("  %s copies=%d %-36s rel-L2 %.3e max-rel %.3e   (reference's own bf16 autocast: %.3e / %.3e)" % (mode, copies, k, e2, em, r2, rm))
        if e2 > lim2 or em > limm:
            bad.append((k, e2, em, lim2, limm))
    enc = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for n_, p in m.named_parameters()
                      if (n_.startswith("encoder.") or n_.startswith("quant_conv.")) and p.grad is not None)) / copies
    print("encoder backward under the reference dz (%s, copies=%d): gradient norm %.5f vs %.5f" % (mode, copies, enc, float(g["gradnorm_encoder"])))
    assert not bad, bad
    assert abs(enc - float(g["gradnorm_encoder"])) < (2e-4 if mode == "fp32" else 2e-2) * float(g["gradnorm_encoder"])


# --------------------------------------------------------------------------------------------------------------
# 4. MakeAScene at config 4's width
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32", "autocast_bf16"])
def test_make_a_scene_w1024_vs_reference_golden(golden_dir, mode):
    """reference models/transformer.py:275-378 at hidden 1024 / 16 heads / S=1536 (BASELINE config 4's row operators at their real
    width: LayerNorm fork / sandwich, tanh-GELU on [1536, 4096], Linear + mas_colsum bias gradients, causal attention hd=64,
    the 8192-way logits).  fp32: logits 1e-3 of max|logit|, gradients 5e-3.  bf16 autocast (what bench.py --workload transformer
    times): logits 3e-2, loss 1e-2, gradients 6e-2 of their max (the tiny-model test's tolerances)."""
    sys.path.insert(0, golden_dir)
    from r3_spec import LOGITS_SUB, TR1024, TR_GRADS
    from mas_hip import ops
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "transformer_w1024.npz"))
    if mode == "fp32":
        ops.set_compute_dtype(torch.float32)
    m = MakeAScene(**TR1024)
    m.load_state_dict(TO.synth_transformer_state_dict(TR1024, seed=9), strict=True)
    m = m.to(dev)
    text, seg, img = (t.to(dev) for t in TO.synth_tokens(TR1024, batch=1, seed=9))
    if mode == "fp32":
        logits = m(text, seg, img)
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = m(text, seg, img)
    assert tuple(logits.shape) == (1, 1024, 8192)
    loss = F.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    tol_log, tol_loss, tol_g = (1e-3, 1e-4, 5e-3) if mode == "fp32" else (3e-2, 1e-2, 6e-2)
    e_log = float(np.abs(logits.detach().float().cpu().numpy()[LOGITS_SUB] - g["logits_sub"]).max()) / float(g["logits_absmax"])
    print("MakeAScene d=1024 %s: logits err %.3e of max|logit|, loss %.5f vs %.5f" % (mode, e_log, float(loss), float(g["loss"])))
    assert e_log < tol_log
    assert abs(float(loss) - float(g["loss"])) < tol_loss * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    for k, sl in TR_GRADS.items():
        got = params[k].grad.detach().float().cpu()[sl]
        e = relerr(got, g["grad:" + k])
        print("  grad %-48s max-rel %.3e" % (k, e))
        assert params[k].grad.dtype == torch.float32 and e < tol_g, k


# --------------------------------------------------------------------------------------------------------------
# 5. the persistent weight-gradient scratch (mas_wgrad_commit)
# --------------------------------------------------------------------------------------------------------------
def test_wgrad_of_every_geometry_is_bitwise_reproducible_and_the_scratch_stays_zero():
    """Round 6: every weight gradient is split-K SLABS + a fixed-order reduce now -- the shapes without a kernel of their own (here: 1x1 at
    64 -> 128 channels, 4x4 stride 2 through space-to-depth, 3x3 at 96 -> 160 channels, and 3x3 in fp32) take the general kernels of
    conv_wgrad.hip in slab mode instead of their fp32 atomics (mas_conv_wgrad_splits > 0 for all of them).  Against autograd of F.conv2d on
    the CPU, and two calls bitwise equal (weights AND bias: the in-work-group bias sums went through LDS float atomics before).  The zeroed
    scratch of the atomic route (still there behind MAS_WGRAD_GENERAL_SLABS=0 and for channel counts the reduce cannot read) stays all-zero."""
    from mas_hip import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(7)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    for (n, cin, h, cout, ks, stride, pad, dt) in ((4, 128, 32, 128, 3, 1, 1, torch.bfloat16), (4, 64, 16, 128, 1, 1, 0, torch.bfloat16),
                                                   (4, 64, 32, 128, 4, 2, 1, torch.bfloat16), (2, 256, 32, 256, 3, 1, 1, torch.bfloat16),
                                                   (3, 96, 24, 160, 3, 1, 1, torch.bfloat16), (2, 64, 16, 128, 4, 1, 1, torch.bfloat16),
                                                   (2, 32, 20, 64, 3, 1, 1, torch.float32), (2, 64, 16, 64, 3, 2, 1, torch.float32)):
        x = torch.randn(n, cin, h, h, generator=g).to(dt)
        ho = (h + 2 * pad - ks) // stride + 1
        dy = torch.randn(n, cout, ho, ho, generator=g).to(dt)
        xd, dyd = cl(x), cl(dy)
        dw, db = ops.conv_wgrad_raw(xd, None, dyd, n, h, h, cin, ho, ho, cout, ks, stride, pad, pad, 0, False, True)
        dw2, db2 = ops.conv_wgrad_raw(xd, None, dyd, n, h, h, cin, ho, ho, cout, ks, stride, pad, pad, 0, False, True)
        torch.cuda.synchronize()
        wr = torch.zeros(cout, cin, ks, ks, requires_grad=True)
        F.conv2d(x.float(), wr, None, stride=stride, padding=pad).backward(dy.float())
        assert dw.shape == wr.shape and dw.is_contiguous()
        tol = 2e-3 if dt == torch.bfloat16 else 2e-5
        assert relerr(dw, wr.grad) < tol and relerr(db, dy.float().sum((0, 2, 3))) < tol, (n, cin, h, cout, ks, stride, dt)
        assert torch.equal(dw, dw2) and torch.equal(db, db2), (n, cin, h, cout, ks, stride, dt)
        for acc in ops._wgrad_scratch.values():
            assert float(acc.abs().max()) == 0.0


def test_split_k_partials_are_deterministic_and_need_no_zeroing():
    """The 3x3 stride-1 weight gradients that carry the FLOPs store split-K PARTIALS (mas_conv_wgrad_partial) that mas_wgrad_reduce adds
    in a fixed order -- no fp32 atomics.  (1) against autograd of F.conv2d on the CPU, ragged tile edges / upsample fold / activation
    prologue included; (2) the workspace is poisoned with NaN before each call: every element a reduce reads was written by the launch
    before it; (3) two runs are BITWISE equal (the atomic commit was not); (4) the atomic path (MAS_WGRAD_PARTIALS=0 route, reached
    here through mas_conv_wgrad) agrees to fp32 summation-order noise."""
    import ctypes as C
    import mas_hip
    from mas_hip import ops, ACT_NONE, ACT_AFFINE_SILU
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    for (n, cin, h, w, cout, act, up) in ((4, 128, 32, 32, 128, ACT_NONE, False), (3, 64, 24, 40, 256, ACT_AFFINE_SILU, False),
                                          (2, 128, 12, 20, 128, ACT_NONE, True), (16, 512, 16, 16, 512, ACT_NONE, False)):
        x = torch.randn(n, cin, h, w, generator=g).bfloat16()
        ho, wo = (2 * h, 2 * w) if up else (h, w)
        dy = (0.25 * torch.randn(n, cout, ho, wo, generator=g)).bfloat16()
        ss = None
        if act != ACT_NONE:
            ss = torch.stack([1.0 + 0.2 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)], dim=-1).contiguous()
        d = ops._desc(n, h, w, cin, ho, wo, cout, 3, 1, 1, 1, torch.bfloat16, torch.bfloat16, act, up)
        ns = mas_hip.lib().mas_conv_wgrad_splits(C.byref(d))
        assert ns > 0, "this shape must take the split-K partial path"
        xd, dyd, ssd = cl(x), cl(dy), (ss.to(dev) if ss is not None else None)
        outs = []
        for _ in range(2):
            for ws in ops._wgrad_partials.values():
                ws.fill_(float("nan"))
            dw, db = ops.conv_wgrad_raw(xd, ssd, dyd, n, h, w, cin, ho, wo, cout, 3, 1, 1, 1, act, up, True)
            torch.cuda.synchronize()
            outs.append((dw.clone(), db.clone()))
        assert len(ops._wgrad_partials) >= 1
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "split-K reduce is not run-to-run bitwise"
        a = x.float()
        if ss is not None:
            a = _silu(a * ss[..., 0][:, :, None, None] + ss[..., 1][:, :, None, None]).bfloat16().float()
        if up:
            a = F.interpolate(a, scale_factor=2.0, mode="nearest")
        wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        F.conv2d(a, wr, None, padding=1).backward(dy.float())
        assert outs[0][0].shape == wr.shape and outs[0][0].is_contiguous()
        assert relerr(outs[0][0], wr.grad) < 2e-3 and relerr(outs[0][1], dy.float().sum((0, 2, 3))) < 2e-3, (n, cin, h, w, cout, act, up)
        acc = torch.zeros(cout * 9 * cin + cout, dtype=torch.float32, device=dev)
        mas_hip.check(mas_hip.lib().mas_conv_wgrad(C.byref(d), ops._ptr(xd), ops._ptr(ssd), ops._ptr(dyd), ops._ptr(acc),
                                                   C.c_void_p(acc.data_ptr() + 4 * cout * 9 * cin), ops._stream()), "conv_wgrad")
        torch.cuda.synchronize()
        dwa = acc[:cout * 9 * cin].view(cout, 3, 3, cin).permute(0, 3, 1, 2)
        assert relerr(outs[0][0], dwa) < 1e-5 and relerr(outs[0][1], acc[cout * 9 * cin:]) < 1e-5


# --------------------------------------------------------------------------------------------------------------
# 6. materialised GroupNorm(+SiLU) output (mas_gn_act) == what the fused loaders form
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(3, 128, 40, 24), (2, 32, 9, 7), (4, 512, 16, 16)])
def test_gn_act_equals_the_fused_prologue(shape, dtype):
    """a = act(x * scale + shift) written by mas_gn_act (reference models/modules.py:121-128 as a tensor) against fp32 torch on the
    CPU, and -- the property the training path relies on -- conv(a) with no prologue == conv(x) with the fused prologue, bitwise:
    both round the activated operand to the activation dtype the same way."""
    from mas_hip import ops, ACT_AFFINE, ACT_AFFINE_SILU, ACT_NONE
    dev = _dev()
    n, c, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + c)
    x = torch.randn(n, c, h, w, generator=g).to(dtype)
    ss = torch.stack([1.0 + 0.2 * torch.randn(n, c, generator=g), 0.3 * torch.randn(n, c, generator=g)], dim=-1).contiguous()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    for act in (ACT_AFFINE_SILU, ACT_AFFINE):
        u = x.float() * ss[..., 0][:, :, None, None] + ss[..., 1][:, :, None, None]
        ref = (_silu(u) if act == ACT_AFFINE_SILU else u).to(dtype).float()
        a = ops.gn_act(xd, ss.to(dev), act)
        assert a.dtype == dtype and a.is_contiguous(memory_format=torch.channels_last)
        e = relerr(a, ref)
        assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), (shape, dtype, act, e)
        if c % 64 == 0 or dtype == torch.float32:
            wt = (torch.randn(64, c, 3, 3, generator=g) / (9 * c) ** 0.5).to(dev)
            y_fused = ops.conv_fwd_raw(xd, ss.to(dev), ops.ConvWeight(wt, False), None, None, n, h, w, c, h, w, 64, 3, 1, 1, 1, act, False, dtype)
            y_mat = ops.conv_fwd_raw(a, None, ops.ConvWeight(wt, False), None, None, n, h, w, c, h, w, 64, 3, 1, 1, 1, ACT_NONE, False, dtype)
            assert torch.equal(y_fused, y_mat), (shape, dtype, act)


# --------------------------------------------------------------------------------------------------------------
# 6b. 1x1 convolutions on the GEMM kernel (conv1x1.hip)
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 128, 16, 16), (3, 192, 384, 9, 7), (32, 512, 1536, 16, 16), (1, 128, 256, 40, 24), (4, 256, 128, 5, 5),
                                   (2, 320, 128, 13, 11), (1, 448, 256, 33, 9)])      # (5 / 7 chunks)
def test_conv1x1_gemm_kernel_vs_cpu_fp32(shape):
    """nin_shortcut / AttnBlock q,k,v,proj_out (reference models/modules.py:106-108,145-160) as plain GEMMs: forward with bias and
    residual, the data gradient (transposed weight image) and the weight gradient, against F.conv2d / autograd in fp32 on the CPU.
    Ragged pixel counts (not a multiple of the 128-pixel tile), 1 / 3 / 8 / 24 channel chunks (odd and even: both LDS stages end the
    loop), several cout tiles."""
    from mas_hip import ops
    dev = _dev()
    n, cin, cout, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
    b = 0.1 * torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g).bfloat16()
    dy = torch.randn(n, cout, h, w, generator=g).bfloat16()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = torch.nn.Parameter(wt.to(dev))
    bd = torch.nn.Parameter(b.to(dev))
    rd = res.to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.norm_act_conv(xd, wd, bd, residual=rd, padding=(0, 0, 0, 0))
    y.backward(dy.to(dev).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    xr = x.float().requires_grad_(True)
    wr = wt.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br) + res.float()
    yr.backward(dy.float())
    assert y.shape == yr.shape and relerr(y.float(), yr) < 1e-2, shape
    assert relerr(xd.grad.float(), xr.grad) < 1e-2 and relerr(wd.grad, wr.grad) < 2e-3 and relerr(bd.grad, br.grad) < 2e-3, shape


# --------------------------------------------------------------------------------------------------------------
# 6c. the RGB-edge layers' weight gradients on the thin transpose-read kernel (conv_thin.hip)
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 128, 8, 10, 24), (2, 8, 128, 10, 24), (3, 128, 8, 32, 32), (3, 8, 128, 33, 17), (32, 8, 128, 64, 64)])
def test_thin_wgrad_kernel_vs_cpu_fp32(shape):
    """conv_out (128 -> 3, zero-padded to 8) and conv_in (3 -> 8 -> 128) weight / bias gradients (autograd of reference
    models/modules.py:219,345): ragged tiles in both directions (the 4 x 16-pixel tile against 10 x 24 / 33 x 17 maps), image borders
    (the zero padding of the convolution = out-of-range halo pixels), several tiles per work-group; against autograd of F.conv2d in
    fp32 on the CPU, and bitwise run to run."""
    import ctypes as C
    import mas_hip
    from mas_hip import ops, ACT_NONE
    dev = _dev()
    n, cin, cout, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(cin + 3 * h)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    dy = (0.5 * torch.randn(n, cout, h, w, generator=g)).bfloat16()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    d = ops._desc(n, h, w, cin, h, w, cout, 3, 1, 1, 1, torch.bfloat16, torch.bfloat16, ACT_NONE, False)
    assert mas_hip.lib().mas_conv_wgrad_splits(C.byref(d)) > 0, "this shape must take the thin split-K path"
    outs = []
    for _ in range(2):
        for ws in ops._wgrad_partials.values():
            ws.fill_(float("nan"))
        dw, db = ops.conv_wgrad_raw(cl(x), None, cl(dy), n, h, w, cin, h, w, cout, 3, 1, 1, 1, ACT_NONE, False, True)
        torch.cuda.synchronize()
        outs.append((dw.clone(), db.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(x.float(), wr, None, padding=1).backward(dy.float())
    assert outs[0][0].shape == wr.shape
    assert relerr(outs[0][0], wr.grad) < 2e-3 and relerr(outs[0][1], dy.float().sum((0, 2, 3))) < 2e-3, shape


@pytest.mark.parametrize("shape", [(2, 10, 24), (3, 33, 17), (1, 64, 96), (32, 64, 64)])
def test_thin_forward_kernel_vs_cpu_fp32(shape):
    """conv_in's forward (RGB zero-padded to 8 channels -> 128) and conv_out's data gradient (the same geometry with the transposed,
    flipped weight image) on conv_thin_fwd_kernel: ragged 8 x 32 tiles, image borders, more tiles than work-groups is not needed for a
    non-pipelined epilogue but (32, 64, 64) walks 4 tiles per work-group-slot pair; through the public op (ops.norm_act_conv pads 3 -> 8),
    forward and backward, against F.conv2d / autograd in fp32 on the CPU."""
    from mas_hip import ops
    dev = _dev()
    n, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(h * 31 + w)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    # conv_in: 3 -> 128 with bias
    x = torch.randn(n, 3, h, w, generator=g).bfloat16()
    w1 = 0.2 * torch.randn(128, 3, 3, 3, generator=g)
    b1 = 0.1 * torch.randn(128, generator=g)
    dy = torch.randn(n, 128, h, w, generator=g).bfloat16()
    w1d, b1d = torch.nn.Parameter(w1.to(dev)), torch.nn.Parameter(b1.to(dev))
    y = ops.norm_act_conv(cl(x), w1d, b1d)
    y.backward(cl(dy))
    w1r, b1r = w1.bfloat16().float().requires_grad_(True), b1.clone().requires_grad_(True)
    yr = F.conv2d(x.float(), w1r, b1r, padding=1)
    yr.backward(dy.float())
    assert relerr(y.float(), yr) < 1e-2 and relerr(w1d.grad, w1r.grad) < 2e-3 and relerr(b1d.grad, b1r.grad) < 2e-3, shape
    # conv_out: 128 -> 3; its data gradient is the 8 -> 128 geometry
    a = torch.randn(n, 128, h, w, generator=g).bfloat16()
    w2 = 0.05 * torch.randn(3, 128, 3, 3, generator=g)
    b2 = 0.1 * torch.randn(3, generator=g)
    d3 = torch.randn(n, 3, h, w, generator=g).bfloat16()
    ad = cl(a).requires_grad_(True)
    w2d, b2d = torch.nn.Parameter(w2.to(dev)), torch.nn.Parameter(b2.to(dev))
    o = ops.norm_act_conv(ad, w2d, b2d)
    o.backward(cl(d3))
    ar = a.float().requires_grad_(True)
    w2r, b2r = w2.bfloat16().float().requires_grad_(True), b2.clone().requires_grad_(True)
    orr = F.conv2d(ar, w2r, b2r, padding=1)
    orr.backward(d3.float())
    torch.cuda.synchronize()
    assert o.shape == orr.shape and relerr(o.float(), orr) < 1e-2, shape
    assert relerr(ad.grad.float(), ar.grad) < 1e-2 and relerr(w2d.grad, w2r.grad) < 2e-3 and relerr(b2d.grad, b2r.grad) < 2e-3, shape


# --------------------------------------------------------------------------------------------------------------
# 6d. Downsample's weight gradient on the stride-2 transpose-read kernel (conv_s2.hip)
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 128, 128, 20, 36), (3, 64, 128, 17, 33), (4, 256, 256, 32, 32), (32, 128, 128, 64, 64), (1, 192, 384, 8, 8)])
def test_stride2_wgrad_kernel_vs_cpu_fp32(shape):
    """reference models/modules.py:62-81 (pad right / bottom by one, 3x3 stride 2, no other padding): weight and bias gradient against
    autograd of F.conv2d on the CPU.  Even and odd map sizes (odd: the padded row / column is never read; even: it is the last tap),
    ragged 4 x 16 output tiles, several cout / cin blocks, many tiles per work-group; bitwise run to run."""
    import ctypes as C
    import mas_hip
    from mas_hip import ops, ACT_NONE
    dev = _dev()
    n, cin, cout, h, w = shape
    ho, wo = (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1
    g = torch.Generator(device="cpu").manual_seed(cin + 5 * h + w)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    dy = (0.5 * torch.randn(n, cout, ho, wo, generator=g)).bfloat16()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    d = ops._desc(n, h, w, cin, ho, wo, cout, 3, 2, 0, 0, torch.bfloat16, torch.bfloat16, ACT_NONE, False)
    assert mas_hip.lib().mas_conv_wgrad_splits(C.byref(d)) > 0, "this shape must take the stride-2 split-K path"
    outs = []
    for _ in range(2):
        for ws in ops._wgrad_partials.values():
            ws.fill_(float("nan"))
        dw, db = ops.conv_wgrad_raw(cl(x), None, cl(dy), n, h, w, cin, ho, wo, cout, 3, 2, 0, 0, ACT_NONE, False, True)
        torch.cuda.synchronize()
        outs.append((dw.clone(), db.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), wr, None, stride=2).backward(dy.float())
    assert outs[0][0].shape == wr.shape
    assert relerr(outs[0][0], wr.grad) < 2e-3 and relerr(outs[0][1], dy.float().sum((0, 2, 3))) < 2e-3, shape


@pytest.mark.parametrize("shape", [(2, 128, 128, 20, 80), (3, 64, 256, 33, 65), (2, 96, 128, 70, 66), (32, 128, 128, 64, 64), (1, 256, 256, 64, 64),
                                   (2, 128, 224, 33, 71), (1, 128, 64, 130, 128),
                                   (32, 256, 256, 32, 32), (3, 128, 128, 37, 35), (2, 128, 128, 24, 30)])   # round 4: half-wide tiles (16 <= Wo < 32); Wo = 15 stays generic
def test_stride2_forward_kernel_vs_cpu_fp32(shape):
    """Downsample's forward (reference models/modules.py:76-79: pad right / bottom by one, 3x3, stride 2) on conv_s2_fwd_kernel: even and
    odd map sizes, ragged 8 x 32 output tiles, 32-channel chunk counts 2 / 3 / 4 / 8 (odd: the low half of the last 64-channel weight
    chunk only), two cout tiles; forward with bias and the whole backward through the public op, against F.conv2d / autograd on the CPU.
    The data gradient of the shapes with Cin % 128 == 0 and W >= 64 runs on conv_s2_dgrad_kernel (four parity classes straight from dy;
    odd sizes: the last dx row / column belongs to classes with fewer valid taps), the others on the zero-stuffed stride-1 path."""
    from mas_hip import ops
    dev = _dev()
    n, cin, cout, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(cin + 7 * h + w)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)
    b = 0.1 * torch.randn(cout, generator=g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
    y = ops.norm_act_conv(xd, wd, bd, stride=2, padding=(0, 1, 0, 1))
    xr = x.float().requires_grad_(True)
    wr, br = wt.bfloat16().float().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, br, stride=2)
    dy = torch.randn(yr.shape, generator=g).bfloat16()
    y.backward(dy.to(dev).contiguous(memory_format=torch.channels_last))
    yr.backward(dy.float())
    torch.cuda.synchronize()
    assert y.shape == yr.shape and relerr(y.float(), yr) < 1e-2, shape
    assert relerr(xd.grad.float(), xr.grad) < 1e-2 and relerr(wd.grad, wr.grad) < 2e-3 and relerr(bd.grad, br.grad) < 2e-3, shape


# --------------------------------------------------------------------------------------------------------------
# 7. MAS_WEIGHT_CACHE_CHECK=1: the debugging aid for writes the packed-weight stamp cannot see
# --------------------------------------------------------------------------------------------------------------
def test_weight_cache_check_flags_a_write_through_data():
    """``w.data.mul_(2)`` changes neither ``_version`` nor the storage: the packed image stays stale until invalidate_weight_cache().
    With MAS_WEIGHT_CACHE_CHECK=1 the next cache hit must raise instead of convolving with the old weights (ops.py:_PackCache)."""
    import subprocess
    _dev()
    env = dict(os.environ, MAS_WEIGHT_CACHE_CHECK="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "helpers", "cache_check.py")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "flagged:" in r.stdout and "after invalidate ok True" in r.stdout, r.stdout
