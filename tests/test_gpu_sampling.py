"""SURVEY 8(f) rank 3 on the MI355X: the decode-time attention kernel (``mas_attn_decode``) against a torch fp32 softmax(qK^T)V,
and KV-cached token-by-token decoding of ``MakeAScene`` against (a) the uncached forward and (b) the REFERENCE's golden logits
(tests/golden/transformer_tiny.npz) -- teacher-forced cached decoding must reproduce them step by step."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relerr(got, ref):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).float()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("case", [
    # B, H, hd, past, nq, S_max, dtype
    (2, 4, 64, 0, 5, 5, torch.float32),          # pure block-causal (prefill semantics) through the decode kernel
    (2, 16, 64, 511, 1, 1536, torch.bfloat16),   # the first decode step of BASELINE config 4 (512 prompt tokens)
    (2, 16, 64, 1534, 1, 1536, torch.bfloat16),  # its last step: 1535 cached rows
    (1, 2, 128, 300, 3, 400, torch.bfloat16),
    (3, 2, 16, 17, 2, 64, torch.float32),
    (2, 2, 32, 1000, 1, 1024, torch.float32),    # more keys than one pass of the 256 lanes, ragged tail
    (1, 1, 64, 0, 1, 8, torch.float32),          # a single key: three of the four waves have none
])
def test_attn_decode_vs_torch(case):
    from mas_hip import ops
    b, h, hd, past, nq, smax, dt = case
    dev = _dev()
    d = h * hd
    g = torch.Generator().manual_seed(past * 7 + nq)
    q = torch.randn(b, nq, d, generator=g).to(dt)
    kc = torch.randn(b, smax, d, generator=g).to(dt)
    vc = torch.randn(b, smax, d, generator=g).to(dt)
    kc[:, past + nq:] = float("nan")             # rows past the valid length must never be read
    vc[:, past + nq:] = float("nan")
    ref = torch.empty(b, nq, d)
    for i in range(nq):
        L = past + i + 1
        qq = q[:, i].float().view(b, h, 1, hd) / math.sqrt(hd)
        k = kc[:, :L].float().view(b, L, h, hd).permute(0, 2, 1, 3)
        v = vc[:, :L].float().view(b, L, h, hd).permute(0, 2, 1, 3)
        ref[:, i] = (torch.softmax(qq @ k.transpose(-1, -2), -1) @ v).reshape(b, d)
    out = ops.attention_decode(q.to(dev), kc.to(dev), vc.to(dev), past, h)
    assert out.dtype == dt and out.shape == (b, nq, d)
    assert relerr(out, ref) < (2e-5 if dt == torch.float32 else 1e-2)
    # a strided view of a fused qkv projection as the query (what SelfAttention passes)
    qkv = torch.randn(b, nq, 3 * d, generator=g).to(dt)
    qkv[..., :d] = q
    out2 = ops.attention_decode(qkv.to(dev)[..., :d], kc.to(dev), vc.to(dev), past, h)
    assert torch.equal(out2, out)


def _golden_model(golden_dir, dev):
    from mas_hip import ops
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    cfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40, text_vocab_size=58,
               image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    m = MakeAScene(**cfg)
    m.load_state_dict(TO.synth_transformer_state_dict(cfg, seed=5), strict=True)
    text, seg, img = (t.to(dev) for t in TO.synth_tokens(cfg, batch=2, seed=5))
    return m.to(dev).eval(), text, seg, img, np.load(os.path.join(golden_dir, "transformer_tiny.npz"))


def test_cached_decoding_vs_uncached_forward_and_reference_golden(golden_dir):
    """teacher-forced cached decoding == the uncached forward == the reference's own logits (fp32 kernels)"""
    dev = _dev()
    m, text, seg, img, g = _golden_model(golden_dir, dev)
    with torch.no_grad():
        full = m(text, seg, img)
        toks, logits = m.generate(text, seg, img_tokens=img, return_logits=True)
    assert torch.equal(toks, img) and logits.shape == (2, 16, 96)
    assert relerr(logits, full.cpu()) < 1e-4
    assert relerr(logits, g["logits"]) < 1e-3
    print(f"cached decode vs uncached forward {relerr(logits, full.cpu()):.2e}, vs reference golden {relerr(logits, g['logits']):.2e}")


def test_cached_decoding_under_autocast_bf16(golden_dir):
    dev = _dev()
    m, text, seg, img, g = _golden_model(golden_dir, dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        toks, logits = m.generate(text, seg, img_tokens=img, return_logits=True)
    assert relerr(logits, g["logits"]) < 3e-2
    k = m.transformer.layers[0].attn
    assert k.cache_capacity == m.total_length


def test_generate_greedy_is_self_consistent_and_decodes_to_an_image():
    """greedy tokens are the argmax of the uncached forward fed with them; ``VQBASE.decode_code`` turns tokens into pixels
    (reference vqvae.py:31-34 calls the non-existent ``embed_code``; ours routes to ``get_codebook_entry``)."""
    from models import VQBASE
    from models.transformer import MakeAScene
    dev = _dev()
    torch.manual_seed(0)
    m = MakeAScene(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=64, seg_vocab_size=11, text_vocab_size=48,
                   image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    text = torch.randint(1, 40, (3, 8), generator=g).to(dev)
    text[:, 6:] = 0
    seg = torch.randint(0, 11, (3, 4), generator=g).to(dev)
    with torch.no_grad():
        tok = m.generate(text, seg, temperature=0)
        assert tok.shape == (3, 16) and int(tok.max()) < 64
        assert (m(text, seg, tok).argmax(-1) == tok).float().mean() > 0.95      # (exact up to fp32 near-ties)
        a = m.generate(text, seg, temperature=0.9, top_k=8, generator=torch.Generator(device=dev).manual_seed(5))
        b = m.generate(text, seg, temperature=0.9, top_k=8, generator=torch.Generator(device=dev).manual_seed(5))
        assert torch.equal(a, b)
        gd = m.generate(text, seg, temperature=0, cond_scale=3.0)
        assert gd.shape == (3, 16)
    vq = VQBASE(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64], num_res_blocks=1, resolution=16,
                              attn_resolutions=[8], dropout=0.0), n_embed=64, embed_dim=32, init_steps=10, reservoir_size=100).to(dev).eval()
    with torch.no_grad():
        img = vq.decode_code(tok.view(3, 4, 4))
    assert img.shape == (3, 3, 8, 8) and torch.isfinite(img).all()


def test_frozen_vq_tokenisation_roundtrip(tmp_path):
    """SURVEY 8(f) rank 4 on the device: ``encode_to_indices`` returns exactly the indices the forward pass quantises with,
    ``decode_code`` of those tokens is the forward's reconstruction, and the tokens survive the shard file bit for bit."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    import token_data as TD
    from mas_hip import ops
    from models import VQBASE
    from oracle import vq_oracle as O
    dev = _dev()
    cfg = dict(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64, 64], num_res_blocks=1, resolution=32,
                             attn_resolutions=[8], dropout=0.0), n_embed=64, embed_dim=32, init_steps=3000, reservoir_size=12500)
    old = ops.compute_dtype()
    ops.set_compute_dtype(torch.float32)
    try:
        m = VQBASE(**cfg)
        m.load_state_dict(O.synth_state_dict(cfg["ddconfig"], 64, 32, seed=0), strict=True)
        m = m.to(dev).eval()
        x = O.synth_image_batch(3, 3, 32, seed=1).to(dev)
        with pytest.raises(RuntimeError):
            m.train().encode_to_indices(x)
        m.eval()
        got = {}
        h = m.quantize.register_forward_hook(lambda mod, i, o: got.__setitem__("idx", o[2]))
        with torch.no_grad():
            rec, _ = m(x)
        h.remove()
        tok = m.encode_to_indices(x)
        assert tok.shape == (3, 64) and tok.dtype == torch.int64      # 32x32 input, two Downsamples -> 8x8 latent grid
        assert torch.equal(tok.reshape(-1), got["idx"].reshape(-1))
        with torch.no_grad():
            rec2 = m.decode_code(tok.view(3, 8, 8))
        assert relerr(rec2, rec.cpu()) < 1e-5
        it, st = TD.tokenize_batch(m, m, x, x)
        paths = TD.write_token_shards(str(tmp_path), [(it, st, torch.zeros(3, 8, dtype=torch.long))], 64, 64, 100)
        img_t, seg_t, _, _, text_t = TD.TokenDataset(paths)[2]
        assert torch.equal(img_t, tok[2].cpu()) and torch.equal(seg_t, tok[2].cpu()) and int(text_t.sum()) == 0
    finally:
        ops.set_compute_dtype(old)


@pytest.mark.parametrize("hidden,heads", [(96, 4), (100, 5)], ids=["hd24", "hd20"])
def test_generate_with_off_config_head_widths(hidden, heads):
    """round 6: head widths the attention kernels do not have (zero-padded in training / prefill, ATen in cached decoding) -- greedy tokens are
    the argmax of the uncached forward fed with them, and cached logits equal uncached logits."""
    from models.transformer import MakeAScene
    dev = _dev()
    torch.manual_seed(2)
    m = MakeAScene(num_layers=2, hidden_dim=hidden, num_attn_heads=heads, image_vocab_size=64, seg_vocab_size=11, text_vocab_size=48,
                   image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    text = torch.randint(1, 40, (2, 8), generator=g).to(dev)
    seg = torch.randint(0, 11, (2, 4), generator=g).to(dev)
    with torch.no_grad():
        tok = m.generate(text, seg, temperature=0)
        assert tok.shape == (2, 16) and int(tok.max()) < 64
        assert (m(text, seg, tok).argmax(-1) == tok).float().mean() > 0.95
