"""CPU oracle for the autoregressive transformer path  --  TEST INFRASTRUCTURE ONLY.

Functional fp32 restatement (torch CPU ops over a flat ``state_dict``) of
``MakeAScene.forward`` (reference models/transformer.py:349-378) in its training
configuration (no KV cache, ``cogview_pb_relax=True``, ``rudalle_relax=False``,
sandwich layer-norm on, prescale off: the constructor defaults, transformer.py:220-225).
Only tests / smoke / the cpu_baseline leg of bench.py may import it.  Pinned against
the reference's own output in tests/golden/transformer_tiny.npz (make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def gelu_tanh(x: Tensor) -> Tensor:
    """OpenAI tanh-GELU (transformer.py:11-14)."""
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


def layer_norm(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def linear(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def causal_attention_scores(q: Tensor, k: Tensor, mask: Tensor, head_dim: int) -> Tensor:
    """``SelfAttention.calculate_attention`` (transformer.py:44-71), pb-relax branch.

    q,k: [B,H,S,hd]; mask: [B,1,S,S] float {0,1}.  Scores are (q/sqrt(hd)) k^T, masked by
    multiply / -1e4 fill (transformer.py:63), then shifted by the per-(batch,head) max of
    scores/32 (transformer.py:64-70) -- a softmax-invariant shift."""
    s = torch.matmul(q / math.sqrt(head_dim), k.transpose(-1, -2))
    s = mask * s - (1.0 - mask) * 10000.0
    alpha = 32.0
    ss = s / alpha
    mx = ss.detach().reshape(s.shape[0], s.shape[1], -1).max(dim=-1)[0][..., None, None]
    return (ss - mx) * alpha


def self_attention(sd: SD, p: str, x: Tensor, mask: Tensor, n_heads: int) -> Tensor:
    """``SelfAttention.forward`` without cache (transformer.py:73-115)."""
    b, s, d = x.shape
    hd = d // n_heads
    qkv = linear(sd, p + ".qkv", x)
    q, k, v = torch.split(qkv, d, dim=-1)
    sp = lambda t: t.view(b, s, n_heads, hd).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    probs = torch.softmax(causal_attention_scores(q, k, mask, hd), dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, s, d)
    return linear(sd, p + ".out_proj", ctx)


def transformer_layer(sd: SD, p: str, x: Tensor, mask: Tensor, n_heads: int) -> Tensor:
    """``TransformerLayer.forward`` (transformer.py:176-210): pre-LN + sandwich-LN."""
    a = self_attention(sd, p + ".attn", layer_norm(sd, p + ".ln_in", x), mask, n_heads)
    x = x + layer_norm(sd, p + ".first_ln_sandwich", a)
    m = layer_norm(sd, p + ".ln_out", x)
    m = linear(sd, p + ".mlp.lin2", gelu_tanh(linear(sd, p + ".mlp.lin1", m)))
    return x + layer_norm(sd, p + ".second_ln_sandwich", m)


def make_a_scene_forward(sd: SD, cfg: dict, text: Tensor, seg: Tensor, img: Tensor) -> Tensor:
    """``MakeAScene.forward`` (transformer.py:349-378).  Returns logits [B, img_len, V_img]."""
    tl, sp, ip = cfg["text_length"], cfg["seg_tokens_per_dim"], cfg["image_tokens_per_dim"]
    img_len, seg_len = ip * ip, sp * sp
    total = tl + seg_len + img_len
    # pad-token remap: 0 -> unique per-position id from the vocab tail (transformer.py:350-353)
    text_range = torch.arange(tl) + (cfg["text_vocab_size"] - tl)
    text = torch.where(text == 0, text_range, text)
    emb_t = F.embedding(text, sd["text_token_embedding.weight"]) + sd["text_pos_embeddings.weight"][: text.shape[1]]
    ar = torch.arange(seg.shape[-1])
    emb_s = (F.embedding(seg, sd["seg_token_embedding.weight"])
             + sd["seg_row_embeddings.weight"][ar // sp] + sd["seg_col_embeddings.weight"][ar % sp])
    ai = torch.arange(img.shape[-1])
    emb_i = (F.embedding(img, sd["image_token_embedding.weight"])
             + sd["image_row_embeddings.weight"][ai // ip] + sd["image_col_embeddings.weight"][ai % ip])
    x = torch.cat((emb_t, emb_s, emb_i), dim=1)
    # mask: tril with a bidirectional prefix block (transformer.py:366-370) ...
    am = torch.tril(torch.ones(x.shape[0], 1, total, total))
    am[:, :, :-img_len, :-img_len] = 1
    am = am[:, :, : x.shape[1], : x.shape[1]]
    # ... re-multiplied per layer by the tril buffer (transformer.py:260-263) => pure causal
    mask = am * sd["transformer.mask"][: am.shape[2], : am.shape[3]]
    for i in range(cfg["num_layers"]):
        x = transformer_layer(sd, f"transformer.layers.{i}", x, mask, cfg["num_attn_heads"])
    x = layer_norm(sd, "transformer.final_ln", x)
    logits = linear(sd, "to_logits.1", layer_norm(sd, "to_logits.0", x))
    return logits[:, -img_len - 1:-1, :]


def synth_transformer_state_dict(cfg: dict, seed: int = 0) -> SD:
    """numpy-seeded MakeAScene ``state_dict`` with the reference's keys and shapes
    (399 entries at 24 layers incl. buffer ``transformer.mask``)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    d = cfg["hidden_dim"]
    sd: SD = {}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))

    def lin(p, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        sd[p + ".weight"] = t(rs.uniform(-b, b, size=(out_f, in_f)))
        sd[p + ".bias"] = t(rs.uniform(-b, b, size=(out_f,)))

    def ln(p):
        sd[p + ".weight"] = t(1.0 + 0.1 * rs.randn(d))
        sd[p + ".bias"] = t(0.1 * rs.randn(d))

    for i in range(cfg["num_layers"]):
        p = f"transformer.layers.{i}"
        ln(p + ".ln_in"); ln(p + ".ln_out"); ln(p + ".first_ln_sandwich"); ln(p + ".second_ln_sandwich")
        lin(p + ".attn.qkv", 3 * d, d); lin(p + ".attn.out_proj", d, d)
        lin(p + ".mlp.lin1", 4 * d, d); lin(p + ".mlp.lin2", d, 4 * d)
    total = cfg["text_length"] + cfg["seg_tokens_per_dim"] ** 2 + cfg["image_tokens_per_dim"] ** 2
    sd["transformer.mask"] = torch.tril(torch.ones(total, total))
    ln("transformer.final_ln")
    sd["image_token_embedding.weight"] = t(rs.randn(cfg["image_vocab_size"], d))
    sd["seg_token_embedding.weight"] = t(rs.randn(cfg["seg_vocab_size"], d))
    sd["text_token_embedding.weight"] = t(rs.randn(cfg["text_vocab_size"], d))
    for name, n in (("text_pos_embeddings", cfg["text_length"]), ("seg_row_embeddings", cfg["seg_tokens_per_dim"]),
                    ("seg_col_embeddings", cfg["seg_tokens_per_dim"]),
                    ("image_row_embeddings", cfg["image_tokens_per_dim"]),
                    ("image_col_embeddings", cfg["image_tokens_per_dim"])):
        sd[name + ".weight"] = t(0.02 * rs.randn(n, d))
    ln("to_logits.0")
    lin("to_logits.1", cfg["image_vocab_size"], d)
    return sd


def synth_tokens(cfg: dict, batch: int, seed: int = 0):
    """Seeded token triples; text gets a zero-padded tail to exercise the pad remap."""
    import numpy as np
    rs = np.random.RandomState(2000 + seed)
    tl = cfg["text_length"]
    text = rs.randint(1, cfg["text_vocab_size"] - tl, size=(batch, tl))
    for b in range(batch):
        text[b, tl - 1 - (b % max(1, tl // 2)):] = 0
    seg = rs.randint(0, cfg["seg_vocab_size"], size=(batch, cfg["seg_tokens_per_dim"] ** 2))
    img = rs.randint(0, cfg["image_vocab_size"], size=(batch, cfg["image_tokens_per_dim"] ** 2))
    return (torch.from_numpy(text.astype(np.int64)), torch.from_numpy(seg.astype(np.int64)),
            torch.from_numpy(img.astype(np.int64)))
