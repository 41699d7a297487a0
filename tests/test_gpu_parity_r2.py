"""Round-2 parity holes (VERDICT r1 "What's weak" 1-3): the configuration bench.py times (bf16, full img_config depth)
against the reference's own output; the dominant launch at its REAL shape against a CPU fp32 convolution; the autograd
surface the VQGAN loss needs (`torch.autograd.grad(..., last_layer.weight, retain_graph=True)` twice, then `backward()`,
with `requires_grad` toggled as reference utils.py:27-29 / train.py:86-98 does); `get_codebook_entry` / `decode_code`;
the packed-weight cache under `.data` writes."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TINY = dict(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64, 64],
                          num_res_blocks=1, resolution=32, attn_resolutions=[8], dropout=0.0),
            n_embed=64, embed_dim=32, init_steps=3000, reservoir_size=12500)
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


@pytest.fixture(autouse=True)
def _restore_dtype():
    from mas_hip import ops
    old = ops.compute_dtype()
    yield
    ops.set_compute_dtype(old)


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


# --------------------------------------------------------------------------------------------------------------
# 1. the benched precision on the benched depth
# --------------------------------------------------------------------------------------------------------------
def test_img256_bf16_vs_reference_golden(golden_dir):
    """BASELINE config 2's model (conf/img_config.yaml block: 23 + 29 layers, codebook 8192x256) at 256x256, B=1, in the
    PRODUCTION precision (bf16 storage / fp32 accumulate, fp32 latent tail) against the reference's own fp32 CPU output.
    Same split as the tiny-net test (end-to-end indices cannot be bit-exact in bf16): latents, index
    agreement, decoder fed the reference's z_q; plus the end-to-end loss and two gradients.  Measured values are printed
    (DESIGN section 3 quotes them)."""
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_img256.npz"))
    m = _build(IMG, 1, torch.bfloat16)
    x = synth_image_batch(1, 3, 256, seed=1).to(_dev())
    taps = {}
    m.encoder.register_forward_hook(lambda mod, i, o: taps.__setitem__("h", o.detach()))
    m.quant_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: taps.__setitem__("q", o))
    rec, q_loss = m(x)
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    e_h = relerr(taps["h"][:, ::8], g["h_sub"])
    e_z, l2_z = relerr(taps["z"], g["z"]), rel_l2(taps["z"], g["z"])
    agree = float((taps["q"][2].cpu().numpy() == g["idx"]).mean())
    with torch.no_grad():
        rec_ref_zq = m.decode(torch.from_numpy(g["z_q"]).to(_dev()))
    e_dec, l2_dec = relerr(rec_ref_zq[:, :, ::8, ::8], g["rec_sub"]), rel_l2(rec_ref_zq[:, :, ::8, ::8], g["rec_sub"])
    e_rec = relerr(rec[:, :, ::8, ::8], g["rec_sub"])
    params = dict(m.named_parameters())
    e_gd = relerr(params["decoder.model.28.weight"].grad, g["grad:decoder.model.28.weight"])
    e_ge = relerr(params["encoder.model.0.weight"].grad, g["grad:encoder.model.0.weight"])
    tot = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None))
    print("img256 bf16 vs reference fp32: encoder-out max-rel %.3e | z max-rel %.3e rel-L2 %.3e | index agreement %.4f | "
          "decoder(ref z_q) max-rel %.3e rel-L2 %.3e | end-to-end rec max-rel %.3e | loss %.5f vs %.5f | "
          "grad dec.28 %.3e enc.0 %.3e | gradnorm %.4f vs %.4f"
          % (e_h, e_z, l2_z, agree, e_dec, l2_dec, e_rec, float(loss), float(g["loss"]), e_gd, e_ge, tot, float(g["gradnorm_total"])))
    # measured on MI355X (round 2): encoder output max-rel 1.9e-2; z max-rel 3.3e-2, rel-L2 2.7e-2; index agreement 0.945
    # (the REFERENCE run in bf16 agrees with its own fp32 run on 0.904 of the indices); decoder fed the
    # reference z_q max-rel 3.1e-2, rel-L2 2.6e-2.  52 bf16-storage layers deep, these are ~1.5x the 4-level tiny net's.
    assert e_h < 3e-2 and e_z < 5e-2 and l2_z < 4e-2
    assert agree >= 0.93                                # tightened in round 6 to the measured floor (0.945 = 242 / 256) minus margin
    assert e_dec < 5e-2 and l2_dec < 4e-2
    assert abs(float(loss) - float(g["loss"])) < 3e-2 * abs(float(g["loss"]))
    assert e_gd < 1e-1
    # total gradient norm: measured 11.11 vs 11.19 (0.7 %).  The first encoder layer's gradient (e_ge, printed) differs by tens of
    # percent END TO END because ~5 % of the codes differ downstream of the bf16 latents; with the reference's dL/dz injected the
    # same gradient is held to 5e-2 rel-L2 (tests/test_gpu_parity_r3.py::test_img256_bf16_encoder_backward_with_reference_dz)
    assert abs(tot - float(g["gradnorm_total"])) < 2e-2 * float(g["gradnorm_total"])
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


# --------------------------------------------------------------------------------------------------------------
# 2. the dominant launch at its real shape vs a CPU fp32 convolution
# --------------------------------------------------------------------------------------------------------------
def _silu(u):
    return u * torch.sigmoid(u)


def test_stream_conv_real_shape_k64_vs_cpu_fp32():
    """128->128 3x3 at 256x256, bf16, N=2, with a PRE-PACKED K64 weight image: the call therefore stays on the kernels that read
    K64 -- conv3x3_stream.hip (16x16-pixel tiles, 512 of them, one per work-group) -- NOT on the wide kernel bench.py's B=32
    launches take (that one, through the default dispatch and with several tiles per work-group, is
    tests/test_gpu_parity_r3.py::test_dominant_conv_wide_real_shape_vs_cpu_fp32).  Forward (plain and with the GroupNorm+SiLU
    loader), data gradient and weight/bias gradient (plain and with the loader) against F.conv2d / autograd in fp32 on the
    CPU, given the same bf16-rounded operands.  Outputs are bf16: tolerance 1e-2 of max|ref| (0.4 % rounding of the largest
    value + fp32 accumulation order); fp32 weight gradients: 2e-3."""
    from mas_hip import ops
    dev = _dev()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    n, c, h = 2, 128, 256
    g = torch.Generator(device="cpu").manual_seed(21)
    x = torch.randn(n, c, h, h, generator=g).bfloat16()
    w = (torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)).bfloat16().float()
    b = 0.1 * torch.randn(c, generator=g)
    res = torch.randn(n, c, h, h, generator=g).bfloat16()
    dy = torch.randn(n, c, h, h, generator=g).bfloat16()
    ss = torch.stack([1.0 + 0.2 * torch.randn(n, c, generator=g), 0.3 * torch.randn(n, c, generator=g)], dim=-1).contiguous()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    xd, resd, dyd, ssd = cl(x), cl(res), cl(dy), ss.to(dev)
    geo = (h, h, c, h, h, c, 3, 1, 1, 1)
    wp = ops.pack_conv_weight(w.to(dev), False, torch.bfloat16)
    wt = ops.pack_conv_weight(w.to(dev), True, torch.bfloat16)
    xf = x.float()
    # activated operand exactly as the loader forms it: fp32 affine + SiLU of the bf16 input, rounded to bf16
    af = _silu(xf * ss[..., 0][:, :, None, None] + ss[..., 1][:, :, None, None]).bfloat16().float()

    # forward, plain (+bias, +residual)
    y = ops.conv_fwd_raw(xd, None, wp, b.to(dev), resd, n, *geo, 0, False, torch.bfloat16)
    ref = F.conv2d(xf, w, b, padding=1) + res.float()
    e = relerr(y, ref); print("fwd plain BIG tile: %.3e" % e); assert e < 1e-2
    # forward with the GN+SiLU prologue
    y2 = ops.conv_fwd_raw(xd, ssd, wp, b.to(dev), None, n, *geo, 2, False, torch.bfloat16)
    ref2 = F.conv2d(af, w, b, padding=1)
    e = relerr(y2, ref2); print("fwd GN+SiLU loader: %.3e" % e); assert e < 1e-2
    # data gradient (the same kernel on dy with the transposed/flipped image)
    da = ops.conv_fwd_raw(dyd, None, wt, None, None, n, *geo, 0, False, torch.bfloat16)
    refd = F.conv_transpose2d(dy.float(), w, padding=1)
    e = relerr(da, refd); print("dgrad: %.3e" % e); assert e < 1e-2
    # weight / bias gradient, plain and with the loader (conv_wgrad_tr_kernel)
    for act, a_in, s_in in ((0, xf, None), (2, af, ssd)):
        dw, db = ops.conv_wgrad_raw(xd, s_in, dyd, n, *geo, act, False, True)
        wr = torch.zeros(c, c, 3, 3, requires_grad=True)
        F.conv2d(a_in, wr, None, padding=1).backward(dy.float())
        e_w, e_b = relerr(dw, wr.grad), relerr(db, dy.float().sum((0, 2, 3)))
        print("wgrad act=%d: dw %.3e db %.3e" % (act, e_w, e_b))
        assert e_w < 2e-3 and e_b < 2e-3


# --------------------------------------------------------------------------------------------------------------
# 3. the autograd surface of the VQGAN generator step
# --------------------------------------------------------------------------------------------------------------
def _change_requires_grad(params, state):           # reference utils.py:27-29
    for p in params:
        p.requires_grad = state


def test_adaptive_weight_retain_graph_and_requires_grad_toggle():
    """reference losses/loss_img.py:56-66 + train.py:84-98 on our autograd nodes: forward once; requires_grad of every
    model parameter flipped off and on again around the discriminator step; two
    `torch.autograd.grad(loss_i, decoder.model[-1].weight, retain_graph=True)` calls (partial-input gradients through
    every node between the loss and the last layer's weight); then `loss.backward()` through the retained graph.
    The same sequence on the CPU oracle (functional fp32 restatement of the reference) is the check."""
    from oracle import vq_oracle as O
    dev = _dev()
    m = _build(TINY, 0, torch.float32)
    x = O.synth_image_batch(2, 3, 32, seed=0)
    g = torch.Generator(device="cpu").manual_seed(5)
    critic = torch.randn(2, 3, 32, 32, generator=g)              # stand-in for the discriminator's d(logits)/d(rec)

    def sequence(forward, last_w, params, xx, cc):
        rec, q_loss = forward(xx)
        _change_requires_grad(params, False)                      # train.py:86 -- the discriminator step sees detached rec
        d_like = (rec.detach() * cc).mean()
        assert not d_like.requires_grad
        _change_requires_grad(params, True)                       # train.py:89
        nll = (xx - rec).abs().mean()
        g_loss = -(rec * cc).mean()
        nll_g = torch.autograd.grad(nll, last_w, retain_graph=True)[0]
        g_g = torch.autograd.grad(g_loss, last_w, retain_graph=True)[0]
        d_weight = torch.clamp(torch.norm(nll_g) / (torch.norm(g_g) + 1e-4), 0.0, 1e4).detach()
        loss = nll + d_weight * g_loss + q_loss
        loss.backward()
        return rec, nll_g, g_g, d_weight, loss

    params = list(m.parameters())
    last = m.decoder.model[-1].weight
    xd, cd = x.to(dev), critic.to(dev)
    rec, nll_g, g_g, d_w, loss = sequence(lambda t: m(t), last, params, xd, cd)

    sd = O.synth_state_dict(TINY["ddconfig"], TINY["n_embed"], TINY["embed_dim"], seed=0)
    leaves = [v.requires_grad_(True) for v in sd.values() if v.is_floating_point()]
    rrec, rnll_g, rg_g, rd_w, rloss = sequence(lambda t: O.vqbase_forward(sd, t, TINY["ddconfig"], training=True)[:2],
                                               sd["decoder.model.16.weight"], leaves, x, critic)
    assert relerr(rec, rrec) < 2e-3
    assert relerr(nll_g, rnll_g) < 5e-3 and relerr(g_g, rg_g) < 5e-3
    assert abs(float(d_w) - float(rd_w)) < 5e-3 * float(rd_w)
    assert abs(float(loss) - float(rloss)) < 2e-3 * abs(float(rloss))
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in m.named_parameters():
        r = sd[k].grad
        if r is None:
            continue
        assert p.grad is not None, k
        # conv biases in front of a GroupNorm have an analytically zero gradient: both sides hold rounding noise there
        assert float((p.grad.detach().cpu() - r).abs().max()) <= 1e-2 * float(r.abs().max()) + 1e-5 * scale, k


def test_frozen_parameters_skip_weight_gradients():
    """A forward taken while every parameter has requires_grad=False (how the generator step runs the discriminator,
    train.py:91-98) must differentiate w.r.t. the INPUT only: dx equals the oracle's, no parameter receives a .grad."""
    from oracle import vq_oracle as O
    dev = _dev()
    m = _build(TINY, 0, torch.float32)
    _change_requires_grad(m.parameters(), False)
    x = O.synth_image_batch(2, 3, 32, seed=0)
    xd = x.to(dev).requires_grad_(True)
    rec, q = m(xd)
    (rec.square().mean() + q).backward()
    assert all(p.grad is None for p in m.parameters())
    sd = O.synth_state_dict(TINY["ddconfig"], TINY["n_embed"], TINY["embed_dim"], seed=0)
    xr = x.clone().requires_grad_(True)
    rrec, rq = O.vqbase_forward(sd, xr, TINY["ddconfig"], training=True)[:2]
    (rrec.square().mean() + rq).backward()
    assert relerr(xd.grad, xr.grad) < 5e-3


# --------------------------------------------------------------------------------------------------------------
# 4. a10: get_codebook_entry / decode_code
# --------------------------------------------------------------------------------------------------------------
def test_get_codebook_entry_and_decode_code(golden_dir):
    """reference models/modules.py:519-528 (indices -> embedding -> NHWC view -> NCHW) and models/vqvae.py:31-34
    (`decode_code`, whose `embed_code` call is the reference's typo for this lookup): entries bit-exact vs the codebook
    rows, and decode_code(idx) == decode(z_q) for the z_q the forward produced from the same indices."""
    from oracle.vq_oracle import synth_image_batch
    dev = _dev()
    m = _build(TINY, 0, torch.float32, train=False)
    x = synth_image_batch(2, 3, 32, seed=0).to(dev)
    with torch.no_grad():
        z = m.quant_conv(m.encoder(x))
        z_q, _, idx = m.quantize(z)
        b, c, hh, ww = z.shape
        ent = m.quantize.get_codebook_entry(idx.view(b, hh * ww), (b, hh, ww, c))
        assert ent.shape == (b, c, hh, ww)
        cbw = m.quantize.embedding.weight
        assert torch.equal(ent.permute(0, 2, 3, 1).reshape(-1, c), cbw[idx])
        assert torch.equal(m.quantize.get_codebook_entry(idx, None), cbw[idx])          # shape=None branch (modules.py:523)
        assert relerr(ent, z_q) < 1e-6                                                 # forward's z_q is the same gather
        rec_code = m.decode_code(idx.view(b, hh * ww))
        rec_zq = m.decode(ent)
        assert torch.equal(rec_code, rec_zq)
    g = np.load(os.path.join(golden_dir, "vq_tiny.npz"))
    with torch.no_grad():                                                              # the reference's own indices -> its own image
        rec_ref = m.decode_code(torch.from_numpy(g["eval:idx"]).to(dev).view(2, -1))
    assert relerr(rec_ref, g["eval:rec"]) < 2e-3


# --------------------------------------------------------------------------------------------------------------
# 5. packed-weight cache vs `.data` writes
# --------------------------------------------------------------------------------------------------------------
def test_weight_cache_invalidation():
    from mas_hip import ops
    from models.modules import Conv2d
    dev = _dev()
    ops.set_compute_dtype(torch.float32)
    conv = Conv2d(8, 8, 3, 1, 1).to(dev)
    x = torch.randn(1, 8, 8, 8, device=dev)
    y0 = conv(x).detach().clone()
    new_w = torch.randn_like(conv.weight) * 0.1
    conv.weight.data.copy_(new_w)                       # neither _version nor data_ptr changes
    ops.invalidate_weight_cache()                       # the documented remedy
    ref = F.conv2d(x.cpu(), new_w.cpu(), conv.bias.detach().cpu(), padding=1)
    assert relerr(conv(x), ref) < 2e-4 and relerr(y0, ref) > 1e-2
    conv.weight.data.mul_(2.0)
    conv.eval()                                         # a train()/eval() switch drops the cache (EMA swap-in pattern)
    assert relerr(conv(x), F.conv2d(x.cpu(), 2 * new_w.cpu(), conv.bias.detach().cpu(), padding=1)) < 2e-4
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    sd["weight"] = sd["weight"] * 0.5
    conv.load_state_dict(sd)                            # load_state_dict drops it too
    assert relerr(conv(x), F.conv2d(x.cpu(), new_w.cpu(), conv.bias.detach().cpu(), padding=1)) < 2e-4
    with torch.no_grad():
        conv.weight.add_(1.0)                           # ordinary in-place update (what an optimizer does): version bump
    assert relerr(conv(x), F.conv2d(x.cpu(), new_w.cpu() + 1.0, conv.bias.detach().cpu(), padding=1)) < 2e-4


@pytest.mark.parametrize("opt_kw", [dict(fused=True), dict(foreach=True), dict(foreach=False, fused=False), dict(mas=True)])
def test_packed_weights_follow_every_optimizer_flavour(opt_kw):
    """torch.optim.Adam(fused=True) updates parameters WITHOUT bumping Parameter._version (measured); the packed-weight cache and the
    bf16 Linear shadows must still refresh after its step (global optimizer post-step hook): the convolution and the Linear layer after
    one step equal torch's on the UPDATED fp32 parameters, and differ from the outputs before the step."""
    import torch.nn.functional as F
    from mas_hip import ops
    from models.modules import Conv2d
    from models.transformer import Linear
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(11)
    conv = Conv2d(64, 128, 3, 1, 1).cuda()
    lin = Linear(64, 64).cuda()
    x = torch.randn(2, 64, 32, 32, device="cuda")
    xl = torch.randn(7, 64, device="cuda")
    if opt_kw.get("mas"):
        from mas_hip.optim import Adam as MasAdam                  # round 4: the library's own one-launch Adam
        opt = MasAdam(list(conv.parameters()) + list(lin.parameters()), lr=0.05)
    else:
        opt = torch.optim.Adam(list(conv.parameters()) + list(lin.parameters()), lr=0.05, **opt_kw)

    def outs():
        y = conv(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            z = lin(xl)
        return y, z
    y0, z0 = outs()
    (y0.float().square().mean() + z0.float().square().mean()).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    y1, z1 = outs()
    ref_y = F.conv2d(x.bfloat16().float(), conv.weight.detach().bfloat16().float(), conv.bias.detach(), padding=1)
    ref_z = F.linear(xl.bfloat16().float(), lin.weight.detach().bfloat16().float(), lin.bias.detach().bfloat16().float())
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    assert rel(y1, ref_y) < 1e-2 and rel(z1, ref_z) < 1e-2, (rel(y1, ref_y), rel(z1, ref_z))
    assert rel(y0, ref_y) > 5e-2 and rel(z0, ref_z) > 5e-2              # the step really moved the outputs
