"""Host logic of the KV-cached sampling path on CPU: the cache bookkeeping of SelfAttention /
TransformerLayer / Transformer / MakeAScene.generate, with the HIP operators they call replaced by torch-CPU stand-ins
(TEST ONLY -- the product path has no CPU fallback; the real kernels run in tests/test_gpu_sampling.py).  Pins the only
behaviour a cache can have: teacher-forced cached decoding reproduces the logits of the uncached forward, which
tests/test_oracle_golden.py / test_gpu_transformer.py pin to the reference's golden logits."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))


def _cpu_ops(monkeypatch):
    from mas_hip import ops

    def causal_attention(qkv, n_heads, dtype=None):
        b, s, d3 = qkv.shape
        d = d3 // 3
        hd = d // n_heads
        sp = lambda t: t.view(b, s, n_heads, hd).permute(0, 2, 1, 3)
        q, k, v = (sp(t) for t in qkv.split(d, dim=-1))
        sc = (q / math.sqrt(hd)) @ k.transpose(-1, -2)
        sc = sc.masked_fill(~torch.tril(torch.ones(s, s, dtype=torch.bool)), float("-inf"))
        return (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(b, s, d)

    def attention_decode(q, kc, vc, past, n_heads):
        b, nq, d = q.shape
        hd = d // n_heads
        out = torch.empty_like(q)
        for i in range(nq):
            L = past + i + 1
            qq = q[:, i].view(b, n_heads, 1, hd) / math.sqrt(hd)
            k = kc[:, :L].view(b, L, n_heads, hd).permute(0, 2, 1, 3)
            v = vc[:, :L].view(b, L, n_heads, hd).permute(0, 2, 1, 3)
            out[:, i] = (torch.softmax(qq @ k.transpose(-1, -2), -1) @ v).reshape(b, d)
        return out

    def layer_norm(x, w, b_, eps=1e-5, residual=None, out_dtype=None, producer_bias_grad=False):
        y = F.layer_norm(x, (x.shape[-1],), w, b_, eps)
        return y if residual is None else residual + y

    gelu = lambda x: 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))
    monkeypatch.setattr(ops, "causal_attention", causal_attention)
    monkeypatch.setattr(ops, "attention_decode", attention_decode)
    monkeypatch.setattr(ops, "layer_norm", layer_norm)
    monkeypatch.setattr(ops, "layer_norm_fork", lambda x, w, b_, eps=1e-5, out_dtype=None: (layer_norm(x, w, b_, eps), x))
    monkeypatch.setattr(ops, "gelu_tanh", gelu)


def _tiny():
    from models.transformer import MakeAScene
    torch.manual_seed(0)
    m = MakeAScene(num_layers=3, hidden_dim=64, num_attn_heads=4, image_vocab_size=50, seg_vocab_size=11, text_vocab_size=40 + 8,
                   image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8).eval()
    g = torch.Generator().manual_seed(1)
    text = torch.randint(1, 40, (2, 8), generator=g)
    text[:, 5:] = 0
    seg = torch.randint(0, 11, (2, 4), generator=g)
    img = torch.randint(0, 50, (2, 16), generator=g)
    return m, text, seg, img


def test_cached_decoding_equals_uncached_forward(monkeypatch):
    _cpu_ops(monkeypatch)
    m, text, seg, img = _tiny()
    with torch.no_grad():
        full = m(text, seg, img)                                            # [B, 16, 50]
        toks, logits = m.generate(text, seg, img_tokens=img, return_logits=True)
    assert torch.equal(toks, img)
    assert float((logits - full).abs().max() / full.abs().max()) < 1e-5


def test_cache_tuple_shapes_follow_the_reference_convention(monkeypatch):
    """cache[i] = (k, v, attn_out, layer_out); k / v are [B, H, L, hd] with the cached length on axis -2 (transformer.py:75,185)"""
    _cpu_ops(monkeypatch)
    m, text, seg, img = _tiny()
    with torch.no_grad():
        emb = m._prompt_embeddings(text, seg)
        out, cache = m.transformer(emb, None, cache={}, use_cache=True)
        assert out.shape == (2, 12, 64) and len(cache) == 3
        k, v, ao, lo = cache[0]
        assert k.shape == (2, 4, 12, 16) and v.shape == k.shape and ao.shape == (2, 12, 64) and lo.shape == (2, 12, 64)
        nxt = torch.cat([emb, m.image_token_embedding(img[:, :1]) + m.get_image_pos_embeddings(img[:, :1])], dim=1)
        out2, cache = m.transformer(nxt, None, cache=cache, use_cache=True)
        assert out2.shape == (2, 1, 64) and cache[2][0].shape[-2] == 13      # only the new position is returned
        with pytest.raises(RuntimeError):                                     # nothing new to compute
            m.transformer.layers[0].attn(nxt, None, True, cache[0][:3])


def test_generate_sampling_modes(monkeypatch):
    _cpu_ops(monkeypatch)
    m, text, seg, img = _tiny()
    with torch.no_grad():
        greedy = m.generate(text, seg, temperature=0)
        assert greedy.shape == (2, 16) and greedy.dtype == torch.long and int(greedy.max()) < 50
        # greedy tokens are the argmax of the uncached forward's logits when fed back (self-consistency)
        full = m(text, seg, greedy)
        assert torch.equal(full.argmax(-1), greedy)
        a = m.generate(text, seg, temperature=1.0, top_k=5, generator=torch.Generator().manual_seed(3))
        b = m.generate(text, seg, temperature=1.0, top_k=5, generator=torch.Generator().manual_seed(3))
        assert torch.equal(a, b)
        # classifier-free guidance with scale 1 is the conditional stream itself
        g1, lg1 = m.generate(text, seg, cond_scale=1.0, img_tokens=img, return_logits=True)
        _, lg = m.generate(text, seg, img_tokens=img, return_logits=True)
        assert float((lg1 - lg).abs().max()) < 1e-4 * float(lg.abs().max())
        # scale 0 is the text-free stream
        _, lg0 = m.generate(text, seg, cond_scale=0.0, img_tokens=img, return_logits=True)
        _, lgu = m.generate(torch.zeros_like(text), seg, img_tokens=img, return_logits=True)
        assert float((lg0 - lgu).abs().max()) < 1e-4 * float(lgu.abs().max())


def test_cache_entries_that_are_no_longer_views_still_decode_correctly(monkeypatch):
    """a cache entry that was cloned / made contiguous / index_select'ed (beam or batch reorder) has lost its backing
    buffer (``_base`` None, layout [B, H, L, hd]); it must be rebuilt from its contents, not reinterpreted as [B, L, H*hd] --
    and a multi-token extend past twice the capacity must not overflow the layer-output buffer."""
    _cpu_ops(monkeypatch)
    m, text, seg, img = _tiny()
    with torch.no_grad():
        emb = m._prompt_embeddings(text, seg)
        step = lambda toks: m.image_token_embedding(toks) + m.get_image_pos_embeddings(toks)
        ref_out, ref_cache = m.transformer(emb, None, cache={}, use_cache=True)
        nxt = torch.cat([emb, step(img[:, :1])], dim=1)
        ref2, _ = m.transformer(nxt, None, cache=ref_cache, use_cache=True)
        # the same with every cache entry detached from its buffer, in three different ways
        _, cache = m.transformer(emb, None, cache={}, use_cache=True)
        perm = torch.tensor([0, 1])
        broken = {i: (c[0].clone(), c[1].contiguous(), c[2].index_select(0, perm), c[3].clone()) for i, c in cache.items()}
        assert all(t._base is None for c in broken.values() for t in c)
        got2, cache2 = m.transformer(nxt, None, cache=broken, use_cache=True)
        assert float((got2 - ref2).abs().max()) < 1e-5 * float(ref2.abs().max())
        # batch reorder: swapping the two samples' cache entries swaps the outputs
        swap = torch.tensor([1, 0])
        _, cache = m.transformer(emb, None, cache={}, use_cache=True)
        swapped = {i: tuple(t.index_select(0, swap) for t in c) for i, c in cache.items()}
        got3, _ = m.transformer(nxt.index_select(0, swap), None, cache=swapped, use_cache=True)
        assert float((got3 - ref2.index_select(0, swap)).abs().max()) < 1e-5 * float(ref2.abs().max())
        # a multi-token extend far past 2x the capacity (12 -> 28 positions in one call)
        many = torch.cat([emb, step(img)], dim=1)
        full, _ = m.transformer(many, None, cache={}, use_cache=True)
        _, cache = m.transformer(emb, None, cache={}, use_cache=True)
        got4, cache4 = m.transformer(many, None, cache=cache, use_cache=True)
        assert got4.shape == (2, 16, 64) and cache4[0][3].shape == (2, 28, 64)
        assert float((got4 - full[:, 12:]).abs().max()) < 1e-5 * float(full.abs().max())
