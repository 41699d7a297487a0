"""End-to-end parity of the drop-in VQBASE (HIP path) against the golden fixtures produced by the
reference itself and against the CPU oracle.  fp32 compute mode isolates kernel correctness
(tolerances ~1e-3: summation order only); bf16 mode states the production tolerance."""
import os

import numpy as np
import pytest
import torch

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
    ref = torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def _build(cfg, seed, dtype, train=True):
    from models import VQBASE
    from mas_hip import ops
    from oracle.vq_oracle import synth_state_dict
    ops.set_compute_dtype(dtype)
    m = VQBASE(**cfg)
    m.load_state_dict(synth_state_dict(cfg["ddconfig"], cfg["n_embed"], cfg["embed_dim"], seed=seed), strict=True)
    m = m.to(_dev()).train(train)
    m.quantize.q_counter = m.quantize.q_re_end          # steady state: VQ active, no k-means
    return m


@pytest.fixture(autouse=True)
def _restore_dtype():
    from mas_hip import ops
    old = ops.compute_dtype()
    yield
    ops.set_compute_dtype(old)


def test_tiny_fp32_train_vs_reference_golden(golden_dir):
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_tiny.npz"))
    m = _build(TINY, 0, torch.float32)
    x = synth_image_batch(2, 3, 32, seed=0).to(_dev())
    taps = {}
    m.quant_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: taps.__setitem__("q", o))
    rec, q_loss = m(x)
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    assert relerr(taps["z"], g["train:z"]) < 1e-3
    assert np.array_equal(taps["q"][2].cpu().numpy(), g["train:idx"])
    assert relerr(rec, g["train:rec"]) < 2e-3
    assert abs(float(q_loss) - float(g["train:q_loss"])) < 1e-3 * abs(float(g["train:q_loss"]))
    assert abs(float(loss) - float(g["train:loss"])) < 1e-3 * abs(float(g["train:loss"]))
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("train:grad:"):
            assert relerr(params[k[len("train:grad:"):]].grad, g[k]) < 5e-3, k
    tot = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None))
    assert abs(tot - float(g["train:gradnorm_total"])) < 5e-3 * tot


def test_tiny_fp32_eval_vs_reference_golden(golden_dir):
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_tiny.npz"))
    m = _build(TINY, 0, torch.float32, train=False)
    x = synth_image_batch(2, 3, 32, seed=0).to(_dev())
    with torch.no_grad():
        rec, q_loss = m(x)
    assert relerr(rec, g["eval:rec"]) < 2e-3


def test_seg_tiny_fp32_vs_reference_golden(golden_dir):
    """VQ-SEG shaped plumbing case (159 one-hot channels in/out; BASELINE config 1)."""
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_seg_tiny.npz"))
    cfg = dict(TINY, ddconfig=dict(TINY["ddconfig"], in_channels=159, out_channels=159))
    m = _build(cfg, 3, torch.float32)
    x = synth_image_batch(2, 159, 16, seed=3).to(_dev())
    rec, q_loss = m(x)
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    assert relerr(rec, g["train:rec"]) < 2e-3
    assert abs(float(loss) - float(g["train:loss"])) < 1e-3 * abs(float(g["train:loss"]))
    params = dict(m.named_parameters())
    for k in ("encoder.model.0.weight", "decoder.model.16.weight"):
        assert relerr(params[k].grad, g["train:grad:" + k]) < 5e-3, k


def test_tiny_bf16_train_vs_reference_golden(golden_dir):
    """Production precision: bf16 activations/weights, fp32 accumulate, fp32 latent tail.
    A bf16 encoder perturbs z by ~1 %, which legitimately flips near-tie codebook indices (the reference
    itself flips 49/512 when run in bf16), so the two halves are checked separately:
    latents within 3e-2 of max|z|; decoder (fed the reference's own z_q) within 3e-2 of max|rec|."""
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_tiny.npz"))
    m = _build(TINY, 0, torch.bfloat16)
    x = synth_image_batch(2, 3, 32, seed=0).to(_dev())
    taps = {}
    m.quant_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: taps.__setitem__("q", o))
    rec, q_loss = m(x)
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    assert relerr(taps["z"], g["train:z"]) < 3e-2
    agree = (taps["q"][2].cpu().numpy() == g["train:idx"]).mean()
    assert agree > 0.85, agree
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    with torch.no_grad():
        rec_from_ref_zq = m.decode(torch.from_numpy(g["train:z_q"]).to(_dev()))
    assert relerr(rec_from_ref_zq, g["train:rec"]) < 3e-2


def test_img256_fp32_vs_reference_golden(golden_dir):
    """conf/img_config.yaml model block, 256x256, B=1, fwd+bwd, against the reference's own output."""
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_img256.npz"))
    m = _build(IMG, 1, torch.float32)
    x = synth_image_batch(1, 3, 256, seed=1).to(_dev())
    taps = {}
    m.quant_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: taps.__setitem__("q", o))
    rec, q_loss = m(x)
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    assert relerr(taps["z"][:, ::8], g["z_sub"]) < 2e-3
    mism = (taps["q"][2].cpu().numpy() != g["idx"]).mean()
    assert mism <= 2 / 256, mism       # end-to-end indices: fp32 summation-order noise may flip a near-tie
    assert relerr(rec[:, :, ::8, ::8], g["rec_sub"]) < 5e-3
    assert abs(float(loss) - float(g["loss"])) < 2e-3 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    assert relerr(params["decoder.model.28.weight"].grad, g["grad:decoder.model.28.weight"]) < 1e-2
    tot = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None))
    assert abs(tot - float(g["gradnorm_total"])) < 2e-2 * tot


def test_img256_fp32_with_bf16_spatial_attention_vs_reference_golden(golden_dir):
    """VERDICT r5 next #3d: in fp32 mode the AttnBlock core is two library GEMMs + softmax, so none of the tight reference-golden
    tests ran spatial_attn.hip.  Here the exact-fp32 model runs its 7 attention cores on the bf16 kernels (forward AND backward;
    ``ops.force_bf16_spatial_attention``) and is held against the reference's own B=1 output: whatever differs from the fp32 run is
    the kernel's bf16 arithmetic (reference models/modules.py:174-187)."""
    from mas_hip import ops
    from oracle.vq_oracle import synth_image_batch
    g = np.load(os.path.join(golden_dir, "vq_img256.npz"))
    m = _build(IMG, 1, torch.float32)
    x = synth_image_batch(1, 3, 256, seed=1).to(_dev())
    taps = {}
    m.quant_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: taps.__setitem__("q", o))
    old = ops.force_bf16_spatial_attention(True)
    try:
        rec, q_loss = m(x)
        loss = (x - rec).abs().mean() + q_loss
        loss.backward()
        with torch.no_grad():
            rec_ref_zq = m.decode(torch.from_numpy(g["z_q"]).to(_dev()))
    finally:
        ops.force_bf16_spatial_attention(old)
    e_z = relerr(taps["z"][:, ::8], g["z_sub"])
    agree = float((taps["q"][2].cpu().numpy() == g["idx"]).mean())
    e_dec = relerr(rec_ref_zq[:, :, ::8, ::8], g["rec_sub"])
    params = dict(m.named_parameters())
    e_gd = relerr(params["decoder.model.28.weight"].grad, g["grad:decoder.model.28.weight"])
    print("img256 fp32 + bf16 spatial attention vs reference: z max-rel %.3e | index agreement %.4f | decoder(ref z_q) max-rel %.3e | "
          "grad dec.28 %.3e | loss %.5f vs %.5f" % (e_z, agree, e_dec, e_gd, float(loss), float(g["loss"])))
    # the bf16 tolerances of the production-precision golden test (tests/test_gpu_parity_r2.py): only 7 of the ~150 layers round here
    assert e_z < 3e-2 and agree >= 0.93 and e_dec < 3e-2
    assert abs(float(loss) - float(g["loss"])) < 3e-2 * abs(float(g["loss"])) and e_gd < 1e-1
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


SEG_YAML = dict(embed_dim=256, n_embed=256, init_steps=3000, reservoir_size=12500,      # conf/seg_config.yaml:13-32 verbatim
                ddconfig=dict(double_z=False, z_channels=256, resolution=256, in_channels=159, out_ch=159, ch=128,
                              ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0))
SEG_EFF = dict(z_channels=256, in_channels=159, out_channels=3, channels=[128, 128, 128, 256, 512, 512], num_res_blocks=2,
               resolution=256, attn_resolutions=[16], dropout=0.0)


def test_seg128_config1_fp32_vs_reference_golden(golden_dir):
    """BASELINE configs[0]: the VQ-SEG model block of conf/seg_config.yaml passed VERBATIM (ch / ch_mult / out_ch / double_z
    are swallowed by **kwargs exactly as the reference swallows them), 128x128, codebook 256, batch 4, fwd+bwd, against the
    reference's own CPU output."""
    from models import VQBASE
    from mas_hip import ops
    from oracle.vq_oracle import synth_image_batch, synth_state_dict
    g = np.load(os.path.join(golden_dir, "vq_seg128.npz"))
    ops.set_compute_dtype(torch.float32)
    m = VQBASE(**SEG_YAML)
    m.load_state_dict(synth_state_dict(SEG_EFF, 256, 256, seed=4), strict=True)
    m = m.to(_dev()).train()
    m.quantize.q_counter = m.quantize.q_re_end
    x = synth_image_batch(4, 159, 128, seed=4).to(_dev())
    taps = {}
    m.quant_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("z", o.detach()))
    m.quantize.register_forward_hook(lambda mod, i, o: taps.__setitem__("q", o))
    rec, q_loss = m(x)
    assert rec.shape == (4, 3, 128, 128)
    loss = rec.abs().mean() + q_loss
    loss.backward()
    assert relerr(taps["z"][:, ::4], g["z_sub"]) < 2e-3
    mism = (taps["q"][2].cpu().numpy() != g["idx"]).mean()
    assert mism <= 2 / 256, mism
    assert relerr(rec[:, :, ::4, ::4], g["rec_sub"]) < 5e-3
    assert abs(float(loss) - float(g["loss"])) < 2e-3 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    assert relerr(params["decoder.model.28.weight"].grad[:, ::8], g["grad:decoder.model.28.weight"]) < 1e-2
    tot = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None))
    assert abs(tot - float(g["gradnorm_total"])) < 2e-2 * tot


def test_img256_bf16_batch_properties():
    """BASELINE config 2 shapes (bf16, 256x256): size-independent properties -- bitwise run-to-run
    determinism of the forward, per-sample independence of the conv/GN stack (eval mode: BN uses
    running stats; checked on the pre-quantisation latents) and finiteness."""
    from oracle.vq_oracle import synth_image_batch
    m = _build(IMG, 1, torch.bfloat16, train=False)
    x = synth_image_batch(4, 3, 256, seed=2).to(_dev())
    with torch.no_grad():
        h4 = m.quant_conv(m.encoder(x))
        h4b = m.quant_conv(m.encoder(x))
        h1 = m.quant_conv(m.encoder(x[1:2]))
        rec4, _ = m(x)
        rec4b, _ = m(x)
    assert torch.equal(h4, h4b)
    # batch 4 and batch 1 do not take the same kernels any more (the dispatch picks the tile geometry that fills the chip:
    # 16x32-pixel tiles / 32-channel chunks at batch 4, 8x16 / 64-channel at batch 1 -- different fp32 accumulation orders),
    # so per-sample independence is checked to bf16 rounding, not bitwise
    assert float((h4[1:2].float() - h1.float()).abs().max() / h1.float().abs().max()) < 2e-2
    assert torch.equal(rec4, rec4b)
    assert torch.isfinite(rec4).all()


def test_kmeans_reinit_vs_oracle():
    """the k-means behind the codebook re-initialisation (reference modules.py:487-499 ->
    fast_pytorch_kmeans, absent and unpinned: parity is against oracle/kmeans_oracle.py's restatement of its published
    algorithm, from identical initial centroids)."""
    from models.kmeans import kmeans_fit
    from oracle.kmeans_oracle import kmeans_lloyd
    dev = _dev()
    rs = np.random.RandomState(3)
    k, d, per = 48, 32, 40
    centers = 4.0 * rs.randn(k, d)
    pts = (centers[:, None, :] + 0.3 * rs.randn(k, per, d)).reshape(-1, d).astype(np.float32)
    rs.shuffle(pts)
    init = rs.choice(len(pts), size=k, replace=False)
    ref_c, ref_a, ref_it = kmeans_lloyd(pts, init)
    cent, idx, it = kmeans_fit(torch.from_numpy(pts).to(dev), k, init_idx=torch.from_numpy(init).to(dev), return_info=True)
    assert it == ref_it
    assert np.array_equal(idx.cpu().numpy(), ref_a)
    assert relerr(cent, ref_c) < 1e-5
    # an empty cluster's centroid becomes the zero vector (library behaviour): duplicate initial centroid -> the higher index never wins a tie
    init2 = init.copy(); init2[1] = init2[0]
    ref_c2, ref_a2, _ = kmeans_lloyd(pts, init2, max_iter=1)
    cent2, idx2, _ = kmeans_fit(torch.from_numpy(pts).to(dev), k, max_iter=1, init_idx=torch.from_numpy(init2).to(dev), return_info=True)
    assert np.array_equal(idx2.cpu().numpy(), ref_a2) and float(cent2[1].abs().max()) == 0.0 and relerr(cent2, ref_c2) < 1e-5


def test_codebook_warmup_schedule():
    """Codebook.forward's training schedule (reference modules.py:474-499): unquantised pass-through with zero loss and no
    indices before 3*init_steps, reservoir of 10 latents per image capped at reservoir_size, k-means re-init at 3*init_steps."""
    import torch.distributed as dist
    from models.modules import Codebook
    from mas_hip import ops
    dev = _dev()
    ops.set_compute_dtype(torch.float32)
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cb = Codebook(16, 32, beta=0.25, init_steps=2, reservoir_size=50).to(dev).train()      # collect > 2, quantise from 6
        g = torch.Generator(device="cpu").manual_seed(0)
        for step in range(1, 8):
            z = torch.randn(4, 32, 6, 6, generator=g).to(dev)
            before = cb.embedding.weight.detach().clone()
            zq, loss, idx = cb(z)
            assert cb.q_counter == step
            if step < 6:
                assert idx is None and float(loss) == 0.0 and torch.equal(zq, z)
            else:
                assert idx is not None and zq.shape == z.shape and float(loss) > 0.0
            if step > 2:
                assert cb.reservoir.shape == (min(50, 40 * (step - 2)), 32)
            changed = not torch.equal(before, cb.embedding.weight.detach())
            assert changed == (step in (6, 7))        # (q - q_init) % q_re_step == 0 with q_re_step = 1: every step from 6 on
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.parametrize("channels,zc,ed,ne", [([32, 96, 160, 192], 48, 24, 50), ([64, 64, 320], 64, 36, 100)],
                         ids=["32-96-160-192", "64-64-320"])
def test_widths_outside_the_reference_configs_vs_oracle(channels, zc, ed, ne):
    """The drop-in keeps the reference's constructor: ANY ``channels`` list GroupNorm(32, C) accepts must train, not only the 128 / 256 / 512
    of conf/*.yaml (round 6: tests/test_gpu_conv_random.py found the GroupNorm backward rejecting C = 192 / 320).  Tiny VQBASE with widths,
    latent and codebook sizes nobody tuned a kernel for, forward + backward, fp32 mode vs the oracle (reference models/vqvae.py:36-39)."""
    from models import VQBASE
    from mas_hip import ops
    from oracle import vq_oracle as O
    dev = _dev()
    cfg = dict(ddconfig=dict(z_channels=zc, in_channels=3, out_channels=3, channels=channels, num_res_blocks=1, resolution=32,
                             attn_resolutions=[8], dropout=0.0), n_embed=ne, embed_dim=ed, init_steps=3000, reservoir_size=12500)
    sd = O.synth_state_dict(cfg["ddconfig"], ne, ed, seed=3)
    x = O.synth_image_batch(2, 3, 32, seed=3)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref, ref_q, ref_idx, ref_z = O.vqbase_forward(sdr, x, cfg["ddconfig"], training=True)
    ((x - ref).abs().mean() + ref_q).backward()
    old = ops.compute_dtype()
    ops.set_compute_dtype(torch.float32)
    try:
        m = VQBASE(**cfg)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        m.quantize.q_counter = m.quantize.q_re_end
        got = {}
        m.quant_conv.register_forward_hook(lambda mod, i, o: got.__setitem__("z", o.detach()))
        m.quantize.register_forward_hook(lambda mod, i, o: got.__setitem__("q", o))
        rec, q = m(x.to(dev))
        ((x.to(dev) - rec).abs().mean() + q).backward()
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype(old)
    assert relerr(got["z"], ref_z.detach()) < 2e-3
    flips = float((got["q"][2].cpu() != ref_idx).float().mean())
    assert flips <= 0.02, flips
    if flips == 0.0:
        assert relerr(rec, ref.detach()) < 5e-3
        params = dict(m.named_parameters())
        checked = 0
        for k, v in sdr.items():
            if v.is_floating_point() and v.grad is not None and k in params and float(v.grad.abs().max()) > 1e-7:
                e = float((params[k].grad.cpu() - v.grad).norm() / (v.grad.norm() + 1e-30))
                assert e < 2e-2, (k, e)
                checked += 1
        assert checked > 30


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-3), ("bf16", 4e-2)])
def test_odd_input_size_batch_and_attention_placement_vs_oracle(mode, tol):
    """Off-config geometry: a 40 x 56 input (ragged tiles at every level, 5 x 7 latent), batch 3, two ResnetBlocks per level, attention at two
    levels -- one of them on a 20 x 28 map (560 tokens: beyond the spatial-attention kernel's 256, so the GEMM path) --, a codebook of 48
    codes in 40 dimensions.  Forward (+ backward in fp32 mode) vs the oracle."""
    from models import VQBASE
    from mas_hip import ops
    from oracle import vq_oracle as O
    dev = _dev()
    cfg = dict(ddconfig=dict(z_channels=40, in_channels=3, out_channels=3, channels=[32, 64, 64, 96], num_res_blocks=2, resolution=32,
                             attn_resolutions=[16, 8], dropout=0.0), n_embed=48, embed_dim=40, init_steps=3000, reservoir_size=12500)
    sd = O.synth_state_dict(cfg["ddconfig"], 48, 40, seed=8)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(3, 3, 40, 56, generator=g)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    taps = {}
    ref, ref_q, ref_idx, ref_z = O.vqbase_forward(sdr, x, cfg["ddconfig"], training=True, taps=taps)
    ((x - ref).abs().mean() + ref_q).backward()
    old = ops.compute_dtype()
    ops.set_compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    try:
        m = VQBASE(**cfg)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        m.quantize.q_counter = m.quantize.q_re_end
        got = {}
        m.quant_conv.register_forward_hook(lambda mod, i, o: got.__setitem__("z", o.detach()))
        rec, q = m(x.to(dev))
        ((x.to(dev) - rec).abs().mean() + q).backward()
        with torch.no_grad():
            dec = m.decode(taps["z_q"].detach().to(dev))
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype(old)
    assert rec.shape == ref.shape and relerr(got["z"], ref_z.detach()) < tol
    assert relerr(dec, ref.detach()) < tol                       # decoder fed the oracle's z_q: no index flips in the way
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    if mode == "fp32":
        params = dict(m.named_parameters())
        k = "decoder.model.%d.weight" % (len(m.decoder.model) - 1)
        e = float((params[k].grad.cpu() - sdr[k].grad).norm() / (sdr[k].grad.norm() + 1e-30))
        assert e < 2e-2, e


def test_side_stream_weight_gradient_is_the_one_stream_gradient_bit_for_bit():
    """ops issues each layer's weight gradient on a second stream beside its GroupNorm backward (MAS_WGRAD_STREAM).  Same
    kernels, same split-K: every parameter gradient and the input gradient must be bitwise what the one-stream order produces -- a missing
    cross-stream dependency would show here.  Also pins the probe's contract: a bool, cached per device, and a refusal is honoured."""
    from mas_hip import ops
    from oracle.vq_oracle import synth_image_batch
    dev = _dev()
    assert isinstance(ops._streams_overlap(torch.cuda.current_stream(), ops._side_stream()), bool)
    alone, both = ops._streams_overlap.last
    assert alone > 0 and both >= 0.9 * alone                  # two spin kernels never finish faster than one
    if not ops._WGRAD_STREAM:
        pytest.skip("MAS_WGRAD_STREAM=0")
    m = _build(IMG, 3, torch.bfloat16)
    x = synth_image_batch(2, 3, 128, seed=4).to(dev).requires_grad_(True)

    def grads(on):
        old = dict(ops._side_ok)
        ops._side_ok[dev.index] = on
        try:
            m.zero_grad(set_to_none=True)
            x.grad = None
            rec, q = m(x)
            ((x - rec).abs().mean() + q).backward()
            torch.cuda.synchronize()
            return [x.grad.clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
        finally:
            ops._side_ok.clear()
            ops._side_ok.update(old)

    one, two, again = grads(False), grads(True), grads(True)
    assert len(one) == len(two) > 100
    for a, b, c in zip(one, two, again):
        assert torch.equal(a, b) and torch.equal(b, c)
