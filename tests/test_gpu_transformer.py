"""Transformer rows of the hot path: the flash-style causal attention kernel vs the CPU oracle
of ``SelfAttention.calculate_attention`` (mask*s - (1-mask)*1e4, PB-relax shift, softmax), and the drop-in
``MakeAScene`` vs the golden logits / gradients produced by the reference itself."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relerr(got, ref):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


@pytest.fixture(autouse=True)
def _restore_dtype():
    from mas_hip import ops
    old = ops.compute_dtype()
    yield
    ops.set_compute_dtype(old)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 24, 16), (1, 2, 200, 64), (2, 16, 1536, 64), (1, 3, 130, 32), (2, 2, 333, 64), (1, 1, 65, 64)])
def test_causal_attention_vs_oracle(shape, dtype):
    from mas_hip import ops
    from oracle import transformer_oracle as TO
    dev = _dev()
    b, h, s, hd = shape
    d = h * hd
    rs = np.random.RandomState(b * 1000 + s)
    qkv = torch.from_numpy(rs.randn(b, s, 3 * d).astype(np.float32))
    if dtype == torch.bfloat16:
        qkv = qkv.bfloat16().float()
    # oracle: reference formula incl. the multiplicative mask with -1e4 fill and the PB-relax shift (transformer.py:44-71)
    ref_in = qkv.clone().requires_grad_(True)
    q, k, v = (t.view(b, s, h, hd).permute(0, 2, 1, 3) for t in torch.split(ref_in, d, dim=-1))
    mask = torch.tril(torch.ones(s, s))[None, None]
    probs = torch.softmax(TO.causal_attention_scores(q, k, mask, hd), dim=-1)
    ref = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, s, d)
    go = torch.from_numpy(rs.randn(b, s, d).astype(np.float32))
    ref.backward(go)
    ops.set_compute_dtype(dtype)
    x = qkv.clone().to(dev).requires_grad_(True)
    out = ops.causal_attention(x, h)
    out.backward(go.to(dev))
    assert relerr(out, ref) < TOL[dtype]
    assert relerr(x.grad, ref_in.grad) < 2 * TOL[dtype]


def test_make_a_scene_fp32_vs_reference_golden(golden_dir):
    from mas_hip import ops
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "transformer_tiny.npz"))
    cfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40, text_vocab_size=58,
               image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    ops.set_compute_dtype(torch.float32)
    m = MakeAScene(**cfg)
    m.load_state_dict(TO.synth_transformer_state_dict(cfg, seed=5), strict=True)
    m = m.to(dev)
    text, seg, img = (t.to(dev) for t in TO.synth_tokens(cfg, batch=2, seed=5))
    logits = m(text, seg, img)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    assert logits.shape == (2, 16, 96)
    assert relerr(logits, g["logits"]) < 1e-3
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad:"):
            assert relerr(params[k[5:]].grad, g[k]) < 5e-3, k


def test_make_a_scene_bf16_logits_close(golden_dir):
    """production precision of the attention core (bf16 MFMA operands, fp32 softmax): logits within 3e-2 of max|logit|."""
    from mas_hip import ops
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "transformer_tiny.npz"))
    cfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40, text_vocab_size=58,
               image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    ops.set_compute_dtype(torch.bfloat16)
    m = MakeAScene(**cfg)
    m.load_state_dict(TO.synth_transformer_state_dict(cfg, seed=5), strict=True)
    m = m.to(dev)
    text, seg, img = (t.to(dev) for t in TO.synth_tokens(cfg, batch=2, seed=5))
    with torch.no_grad():
        logits = m(text, seg, img)
    assert relerr(logits, g["logits"]) < 3e-2
