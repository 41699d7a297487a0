"""Pins the CPU oracle (oracle/) to outputs of the reference itself (tests/golden/*.npz,
produced by tests/golden/make_golden.py which imports /root/reference unmodified)."""
import os

import numpy as np
import pytest
import torch

from oracle import vq_oracle as O
from oracle import transformer_oracle as TO

TINY = dict(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64, 64],
                          num_res_blocks=1, resolution=32, attn_resolutions=[8], dropout=0.0),
            n_embed=64, embed_dim=32)
IMG_DD = dict(z_channels=256, in_channels=3, out_channels=3, channels=[128, 128, 128, 256, 512, 512],
              num_res_blocks=2, resolution=512, attn_resolutions=[32], dropout=0.0)


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _run_train(cfg, x, seed, grad_keys):
    sd = O.synth_state_dict(cfg["ddconfig"], cfg["n_embed"], cfg["embed_dim"], seed=seed)
    for k, v in sd.items():
        if v.is_floating_point():
            v.requires_grad_(True)
    dec, q_loss, idx, z = O.vqbase_forward(sd, x, cfg["ddconfig"], training=True)
    loss = O.recon_vq_loss(x, dec, q_loss)
    loss.backward()
    return sd, dec, q_loss, idx, z, loss


def test_tiny_train_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "vq_tiny.npz"))
    x = O.synth_image_batch(2, 3, 32, seed=0)
    sd, dec, q_loss, idx, z, loss = _run_train(TINY, x, 0, None)
    assert np.array_equal(idx.numpy(), g["train:idx"])          # indices bit-exact
    _close(z, g["train:z"]); _close(dec, g["train:rec"]); _close(q_loss, g["train:q_loss"])
    _close(loss, g["train:loss"])
    for k in g.files:
        if k.startswith("train:grad:"):
            _close(sd[k[len("train:grad:"):]].grad, g[k], rtol=2e-3, atol=1e-6)
    tot = np.sqrt(sum(float((v.grad.double() ** 2).sum()) for v in sd.values() if v.grad is not None))
    assert abs(tot - float(g["train:gradnorm_total"])) < 1e-4 * tot


def test_tiny_eval_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "vq_tiny.npz"))
    x = O.synth_image_batch(2, 3, 32, seed=0)
    sd = O.synth_state_dict(TINY["ddconfig"], TINY["n_embed"], TINY["embed_dim"], seed=0)
    with torch.no_grad():
        dec, q_loss, idx, z = O.vqbase_forward(sd, x, TINY["ddconfig"], training=False)
    assert np.array_equal(idx.numpy(), g["eval:idx"])
    _close(dec, g["eval:rec"]); _close(z, g["eval:z"])


def test_seg_tiny_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "vq_seg_tiny.npz"))
    cfg = dict(TINY, ddconfig=dict(TINY["ddconfig"], in_channels=159, out_channels=159))
    x = O.synth_image_batch(2, 159, 16, seed=3)
    sd, dec, q_loss, idx, z, loss = _run_train(cfg, x, 3, None)
    assert np.array_equal(idx.numpy(), g["train:idx"])
    _close(dec, g["train:rec"]); _close(loss, g["train:loss"])
    _close(sd["encoder.model.0.weight"].grad, g["train:grad:encoder.model.0.weight"], rtol=2e-3, atol=1e-6)


def test_img256_matches_reference(golden_dir):
    """Full VQ-IMG config (conf/img_config.yaml model block), B=1, fwd+bwd."""
    g = np.load(os.path.join(golden_dir, "vq_img256.npz"))
    cfg = dict(ddconfig=IMG_DD, n_embed=8192, embed_dim=256)
    x = O.synth_image_batch(1, 3, 256, seed=1)
    sd, dec, q_loss, idx, z, loss = _run_train(cfg, x, 1, None)
    assert np.array_equal(idx.numpy(), g["idx"])
    _close(dec[:, :, ::8, ::8], g["rec_sub"], rtol=1e-3, atol=1e-4)
    _close(z[:, ::8], g["z_sub"], rtol=1e-3, atol=1e-4)
    _close(loss, g["loss"])
    _close(sd["decoder.model.28.weight"].grad, g["grad:decoder.model.28.weight"], rtol=5e-3, atol=1e-6)
    tot = np.sqrt(sum(float((v.grad.double() ** 2).sum()) for v in sd.values() if v.grad is not None))
    assert abs(tot - float(g["gradnorm_total"])) < 1e-3 * tot


SEG_EFF = dict(z_channels=256, in_channels=159, out_channels=3, channels=[128, 128, 128, 256, 512, 512], num_res_blocks=2,
               resolution=256, attn_resolutions=[16], dropout=0.0)     # what conf/seg_config.yaml's model block resolves to


def test_seg128_config1_matches_reference(golden_dir):
    """BASELINE configs[0]: VQ-SEG 128x128, codebook 256, batch 4 (conf/seg_config.yaml on the reference's CPU path)."""
    g = np.load(os.path.join(golden_dir, "vq_seg128.npz"))
    x = O.synth_image_batch(4, 159, 128, seed=4)
    sd = O.synth_state_dict(SEG_EFF, 256, 256, seed=4)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    dec, q_loss, idx, z = O.vqbase_forward(sd, x, SEG_EFF, training=True)
    loss = dec.abs().mean() + q_loss               # the YAML's decoder emits 3 channels (out_ch is swallowed): no x - rec here
    loss.backward()
    assert np.array_equal(idx.numpy(), g["idx"])
    _close(dec[:, :, ::4, ::4], g["rec_sub"], rtol=1e-3, atol=1e-4)
    _close(z[:, ::4], g["z_sub"], rtol=1e-3, atol=1e-4)
    _close(loss, g["loss"])
    _close(sd["decoder.model.28.weight"].grad[:, ::8], g["grad:decoder.model.28.weight"], rtol=5e-3, atol=1e-6)
    tot = np.sqrt(sum(float((v.grad.double() ** 2).sum()) for v in sd.values() if v.grad is not None))
    assert abs(tot - float(g["gradnorm_total"])) < 1e-3 * tot


@pytest.mark.parametrize("tag", ["scaled", "default"])
def test_codebook_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"codebook_{tag}.npz"))
    rs = np.random.RandomState(7)
    z = torch.from_numpy(rs.randn(4, 256, 16, 16).astype(np.float32))
    scaled = rs.randn(8192, 256).astype(np.float32)
    default = rs.uniform(-1 / 8192, 1 / 8192, size=(8192, 256)).astype(np.float32)
    cb = torch.from_numpy(scaled if tag == "scaled" else default)
    zq, loss, idx = O.codebook_forward(cb, z)
    assert np.array_equal(idx.numpy(), g["idx"])
    _close(loss, g["loss"]); _close(zq[:, ::16], g["zq_sub"])


def test_transformer_tiny_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "transformer_tiny.npz"))
    cfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40,
               text_vocab_size=58, image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    sd = TO.synth_transformer_state_dict(cfg, seed=5)
    for v in sd.values():
        v.requires_grad_(True)
    text, seg, img = TO.synth_tokens(cfg, batch=2, seed=5)
    logits = TO.make_a_scene_forward(sd, cfg, text, seg, img)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    _close(logits, g["logits"], rtol=1e-4, atol=1e-5)
    _close(loss, g["loss"])
    for k in g.files:
        if k.startswith("grad:"):
            _close(sd[k[5:]].grad, g[k], rtol=2e-3, atol=1e-6)


def test_loss_oracle_vs_reference_discriminator_golden(golden_dir):
    """oracle/loss_oracle.py (PatchGAN forward, hinge / vanilla, BatchNorm train + eval) == the reference's own
    losses/discriminator.py + loss_img.py functions (tests/golden/disc_tiny.npz, made by make_loss_golden.py)"""
    import numpy as np
    import torch
    from oracle import loss_oracle as LO
    g = np.load(os.path.join(golden_dir, "disc_tiny.npz"))
    sd = LO.synth_disc_state_dict(seed=7)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    real = torch.from_numpy(g["real"])
    fake = torch.from_numpy(g["fake"]).requires_grad_(True)
    lr, lf = LO.disc_forward(sd, real, True), LO.disc_forward(sd, fake, True)
    rel = lambda a, b: float(np.abs(a.detach().numpy() - b).max() / max(np.abs(b).max(), 1e-12))
    assert rel(lr, g["logits_real"]) < 1e-5 and rel(lf, g["logits_fake"]) < 1e-5
    assert abs(float(LO.hinge_d_loss(lr, lf)) - float(g["hinge"])) < 1e-6
    assert abs(float(LO.vanilla_d_loss(lr, lf)) - float(g["vanilla"])) < 1e-6
    g_loss = -lf.mean()
    (gin,) = torch.autograd.grad(g_loss, fake, retain_graph=True)
    assert rel(gin, g["grad_fake:g_loss"]) < 1e-4
    LO.hinge_d_loss(lr, lf).backward()
    for k in g.files:
        if k.startswith("grad:"):
            got = sd[k[5:]].grad
            got = got[::8, ::8] if got.numel() > 200000 else got
            assert rel(got, g[k]) < 1e-4, k
    with torch.no_grad():
        assert rel(LO.disc_forward(sd, real, False), g["logits_real_eval"]) < 1e-5
    assert LO.adopt_weight(0.8, 10, 20) == float(g["adopt"][0]) == 0.0 and abs(LO.adopt_weight(0.8, 30, 20) - float(g["adopt"][1])) < 1e-7


def test_lpips_oracle_vs_reference_golden(golden_dir):
    """oracle/lpips_oracle.py against the reference's own LPIPS class (tests/golden/make_lpips_golden.py): distance, input
    gradient, the five feature taps, and the state_dict layout our module must expose."""
    import warnings
    from oracle import lpips_oracle as LO
    g = np.load(os.path.join(golden_dir, "lpips_tiny.npz"))
    assert [str(k) for k in g["keys"]] == LO.expected_keys()
    sd = LO.synth_lpips_state_dict(seed=3)
    real = torch.from_numpy(g["real"])
    fake = torch.from_numpy(g["fake"]).requires_grad_(True)
    out = LO.lpips(sd, real, fake)
    out.sum().backward()
    assert out.shape == (2, 1, 1, 1)
    assert np.abs(out.detach().numpy() - g["out"]).max() <= 1e-6 * np.abs(g["out"]).max()
    assert np.abs(fake.grad.numpy() - g["dfake"]).max() <= 1e-5 * np.abs(g["dfake"]).max()
    feats = LO.vgg_features(sd, (real - sd["scaling_layer.shift"]) / sd["scaling_layer.scale"])
    assert [list(f.shape) for f in feats] == g["feat_shapes"].tolist()
    assert np.allclose([float(f.mean()) for f in feats], g["feat_means"], rtol=1e-5)
    # the product's module: same keys in the same order, strict load of a reference-shaped state_dict (no kernel runs here)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import losses
        m = losses.LPIPS()
    assert list(m.state_dict().keys()) == LO.expected_keys()
    m.load_state_dict(sd, strict=True)
    assert not any(p.requires_grad for p in m.parameters())
    assert isinstance(losses.LPIPSWithObject(), losses.LPIPS)


# ---------------------------------------------------------------------------------------------------------------------
# round-3 fixtures (tests/golden/make_golden_r3.py): the oracle reproduces what the GPU tests will be held to
# ---------------------------------------------------------------------------------------------------------------------
def test_img256_encoder_backward_fixture_matches_oracle(golden_dir):
    """vq_img256_bwd.npz: dL/dz at the output of quant_conv and the encoder gradients of the reference's 256x256 B=1 run."""
    import sys
    sys.path.insert(0, golden_dir)
    from r3_spec import ENC_GRADS
    g = np.load(os.path.join(golden_dir, "vq_img256_bwd.npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    x = O.synth_image_batch(1, 3, 256, seed=1)
    sd = O.synth_state_dict(IMG_DD, 8192, 256, seed=1)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    taps = {}
    dec, q_loss, idx, z = O.vqbase_forward(sd, x, IMG_DD, training=True, taps=taps)
    taps["z"].retain_grad()
    O.recon_vq_loss(x, dec, q_loss).backward()
    dz = taps["z"].grad
    assert float((dz - torch.from_numpy(g["dz"])).abs().max()) < 2e-3 * float(np.abs(g["dz"]).max())
    for k, sl in ENC_GRADS.items():
        got, ref = sd[k].grad.numpy()[sl], g["grad:" + k]
        assert got.shape == ref.shape, k
        assert float(np.abs(got - ref).max()) < 5e-3 * float(np.abs(ref).max()) + 1e-9, k
    enc = np.sqrt(sum(float((v.grad.double() ** 2).sum()) for k, v in sd.items()
                      if (k.startswith("encoder.") or k.startswith("quant_conv.")) and v.grad is not None))
    assert abs(enc - float(g["gradnorm_encoder"])) < 1e-3 * enc


def test_transformer_w1024_fixture_matches_oracle(golden_dir):
    """transformer_w1024.npz: MakeAScene at config 4's width (2 layers, d=1024, 16 heads, S=1536, B=1), logits + loss + gradients."""
    import sys
    sys.path.insert(0, golden_dir)
    from r3_spec import LOGITS_SUB, TR1024, TR_GRADS
    g = np.load(os.path.join(golden_dir, "transformer_w1024.npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = TO.synth_transformer_state_dict(TR1024, seed=9)
    for k, v in sd.items():
        if v.is_floating_point() and k != "transformer.mask":
            v.requires_grad_(True)
    text, seg, img = TO.synth_tokens(TR1024, batch=1, seed=9)
    logits = TO.make_a_scene_forward(sd, TR1024, text, seg, img)
    assert tuple(logits.shape) == (1, 1024, 8192)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    amax = float(g["logits_absmax"])
    assert float(np.abs(logits.detach().numpy()[LOGITS_SUB] - g["logits_sub"]).max()) < 1e-4 * amax
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    for k, sl in TR_GRADS.items():
        got, ref = sd[k].grad.numpy()[sl], g["grad:" + k]
        assert float(np.abs(got - ref).max()) < 2e-3 * float(np.abs(ref).max()) + 1e-12, k
