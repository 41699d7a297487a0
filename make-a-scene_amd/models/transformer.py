"""The autoregressive transformer behind the reference's class surface (reference models/transformer.py):
``SelfAttention`` :17, ``MLP`` :118, ``TransformerLayer`` :142, ``Transformer`` :213, ``MakeAScene`` :275 --
same constructor arguments, submodule names and ``state_dict`` keys (incl. the ``transformer.mask`` buffer).

Training configuration only (no KV cache; pb-relax on, rudalle-relax off, sandwich layer-norm on -- the reference's
defaults): the attention core is the flash-style HIP kernel (``mas_attn_causal_fwd``); the PB-relax shift
(:64-70) subtracts a per-(batch, head) constant before the softmax and the per-layer ``mask * tril`` product (:260-263)
is pure causal, so an online-softmax causal kernel computes the same function.  Linear / LayerNorm /
embeddings are plain library calls (hipBLASLt GEMMs), as SURVEY 2.1 K11 allows."""
import math

import torch
import torch.nn as nn
from torch.nn import functional as F

from mas_hip import ops


def gelu(x):
    """OpenAI tanh-GELU (reference transformer.py:11-14): one fused HIP pass instead of ~9 elementwise launches."""
    return ops.gelu_tanh(x)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters / state_dict keys, HIP forward + backward; ``residual`` fuses the ``x + LN(.)`` of the
    sandwich LayerNorms (reference transformer.py:201-203,207-209)."""

    def forward(self, x, residual=None):
        if not self.elementwise_affine or len(self.normalized_shape) != 1:
            raise NotImplementedError("libmas_hip LayerNorm: affine, over the last dimension")
        return ops.layer_norm(x, self.weight, self.bias, self.eps, residual)


class SelfAttention(nn.Module):
    def __init__(self, hidden_dim, num_attn_heads, attn_dropout_prob, out_dropout_prob, cogview_pb_relax=True, rudalle_relax=False):
        super(SelfAttention, self).__init__()
        self.hidden_dim = hidden_dim
        self.num_attn_heads = num_attn_heads
        self.d = math.sqrt(self.hidden_dim // self.num_attn_heads)
        self.qkv = nn.Linear(hidden_dim, 3 * hidden_dim)
        self.attn_drop = nn.Dropout(attn_dropout_prob)
        self.out_proj = nn.Linear(hidden_dim, hidden_dim)
        self.out_drop = nn.Dropout(out_dropout_prob)
        self.cogview_pb_relax = cogview_pb_relax
        self.rudalle_relax = rudalle_relax

    def forward(self, x, mask, use_cache=False, cache=None):
        if use_cache or cache:
            raise NotImplementedError("KV-cached sampling is SURVEY section 8(f) rank 3 (not built yet)")
        if self.rudalle_relax or (self.training and self.attn_drop.p > 0):
            raise NotImplementedError("rudalle_relax / attention dropout are off the measured path (reference defaults: off)")
        qkv = self.qkv(x)
        context = ops.causal_attention(qkv, self.num_attn_heads)     # `mask` is causal by construction (transformer.py:260-263)
        out = self.out_proj(context)
        return self.out_drop(out), cache


class MLP(nn.Module):
    def __init__(self, hidden_dim, dropout_prob, rudalle_relax=False):
        super(MLP, self).__init__()
        self.lin1 = nn.Linear(hidden_dim, 4 * hidden_dim)
        self.lin2 = nn.Linear(4 * hidden_dim, hidden_dim)
        self.dropout = nn.Dropout(dropout_prob)
        self.rudalle_relax = rudalle_relax

    def forward(self, x):
        if self.rudalle_relax:
            raise NotImplementedError("rudalle_relax is off the measured path")
        return self.dropout(self.lin2(gelu(self.lin1(x))))


class TransformerLayer(nn.Module):
    def __init__(self, hidden_dim, num_attn_heads, attn_dropout_prop, out_dropout_prob, cogview_pb_relax=True,
                 cogview_sandwich_layernorm=True, cogview_layernorm_prescale=False, rudalle_relax=False):
        super().__init__()
        self.cogview_pb_relax = cogview_pb_relax
        self.cogview_sandwich_layernorm = cogview_sandwich_layernorm
        self.cogview_layernorm_prescale = cogview_layernorm_prescale
        self.rudalle_relax = rudalle_relax
        self.ln_in = LayerNorm(hidden_dim, eps=1e-5)
        self.ln_out = LayerNorm(hidden_dim, eps=1e-5)
        if cogview_sandwich_layernorm:
            self.first_ln_sandwich = LayerNorm(hidden_dim, eps=1e-5)
            self.second_ln_sandwich = LayerNorm(hidden_dim, eps=1e-5)
        self.attn = SelfAttention(hidden_dim=hidden_dim, num_attn_heads=num_attn_heads, attn_dropout_prob=attn_dropout_prop,
                                  out_dropout_prob=out_dropout_prob, cogview_pb_relax=cogview_pb_relax, rudalle_relax=rudalle_relax)
        self.mlp = MLP(hidden_dim=hidden_dim, dropout_prob=out_dropout_prob, rudalle_relax=rudalle_relax)

    def _prescale(self, t):
        return t / t.detach().max(dim=-1)[0].unsqueeze(-1) if self.cogview_layernorm_prescale else t

    def forward(self, x, mask, cache=None, use_cache=False, mlp_cache=False):
        attn_out, new_cache = self.attn(self.ln_in(self._prescale(x)), mask, use_cache, cache)
        if self.cogview_sandwich_layernorm:
            x = self.first_ln_sandwich(self._prescale(attn_out), residual=x)        # x + LN(attn_out), one pass
        else:
            x = x + attn_out
        mlp_out = self.mlp(self.ln_out(self._prescale(x)))
        if self.cogview_sandwich_layernorm:
            return self.second_ln_sandwich(mlp_out, residual=x), new_cache
        return x + mlp_out, new_cache


class Transformer(nn.Module):
    def __init__(self, num_layers, hidden_dim, num_attn_heads, image_tokens_per_dim, seg_tokens_per_dim, text_length,
                 attn_dropout_prop=0, out_dropout_prob=0, cogview_pb_relax=True, cogview_sandwich_layernorm=True,
                 cogview_layernorm_prescale=False, rudalle_relax=False):
        super(Transformer, self).__init__()
        self.num_layers = num_layers
        self.cogview_pb_relax = cogview_pb_relax
        self.rudalle_relax = rudalle_relax
        self.layers = nn.ModuleList([
            TransformerLayer(hidden_dim, num_attn_heads, attn_dropout_prop, out_dropout_prob, cogview_pb_relax,
                             cogview_sandwich_layernorm, cogview_layernorm_prescale, rudalle_relax) for _ in range(num_layers)])
        self.register_buffer("mask", self._create_mask(text_length, seg_tokens_per_dim, image_tokens_per_dim))
        self.final_ln = LayerNorm(hidden_dim, eps=1e-5)

    def _create_mask(self, text_length, seg_tokens_per_dim, image_tokens_per_dim):
        size = text_length + seg_tokens_per_dim ** 2 + image_tokens_per_dim ** 2
        return torch.tril(torch.ones(size, size, dtype=torch.float32))

    def forward(self, x, attn_mask, cache=None, use_cache=None):
        if cache is None:
            cache = {}
        for i, layer in enumerate(self.layers):
            # attn_mask * self.mask is pure causal: nothing to materialise for the kernel
            x, layer_cache = layer(x, None, cache.get(i), mlp_cache=i == len(self.layers) - 1, use_cache=use_cache)
            cache[i] = layer_cache
        return self.final_ln(x), cache


class MakeAScene(nn.Module):
    def __init__(self, num_layers, hidden_dim, num_attn_heads, image_vocab_size, seg_vocab_size, text_vocab_size,
                 image_tokens_per_dim, seg_tokens_per_dim, text_length):
        super(MakeAScene, self).__init__()
        self.image_tokens_per_dim = image_tokens_per_dim
        self.seg_tokens_per_dim = seg_tokens_per_dim
        self.image_length = image_tokens_per_dim ** 2
        self.seg_length = seg_tokens_per_dim ** 2
        self.text_length = text_length
        self.total_length = self.text_length + self.seg_length + self.image_length
        self.text_vocab_size = text_vocab_size
        self.transformer = Transformer(num_layers, hidden_dim, num_attn_heads, image_tokens_per_dim, seg_tokens_per_dim, text_length)
        self.image_token_embedding = nn.Embedding(image_vocab_size, hidden_dim)
        self.seg_token_embedding = nn.Embedding(seg_vocab_size, hidden_dim)
        self.text_token_embedding = nn.Embedding(text_vocab_size, hidden_dim)
        self.text_pos_embeddings = torch.nn.Embedding(text_length, hidden_dim)
        self.seg_row_embeddings = torch.nn.Embedding(seg_tokens_per_dim, hidden_dim)
        self.seg_col_embeddings = torch.nn.Embedding(seg_tokens_per_dim, hidden_dim)
        self.image_row_embeddings = torch.nn.Embedding(image_tokens_per_dim, hidden_dim)
        self.image_col_embeddings = torch.nn.Embedding(image_tokens_per_dim, hidden_dim)
        for m in (self.text_pos_embeddings, self.seg_row_embeddings, self.seg_col_embeddings, self.image_row_embeddings,
                  self.image_col_embeddings):
            self._init_weights(m)
        self.to_logits = torch.nn.Sequential(LayerNorm(hidden_dim), torch.nn.Linear(hidden_dim, image_vocab_size))

    @property
    def device(self):
        """the reference reads ``self.device`` but never sets it (transformer.py:332,352; its __main__ assigns it by hand);
        assigning it still works, otherwise it is where the parameters live"""
        return self.__dict__.get("_device", self.image_token_embedding.weight.device)

    @device.setter
    def device(self, value):
        self.__dict__["_device"] = value

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def get_seg_pos_embeddings(self, seg_input_ids):
        n = seg_input_ids.size(-1)
        ids = torch.arange(n, dtype=torch.long, device=self.device)
        return self.seg_row_embeddings((ids // self.seg_tokens_per_dim)[None]) + self.seg_col_embeddings((ids % self.seg_tokens_per_dim)[None])

    def get_image_pos_embeddings(self, image_input_ids, past_length=0):
        n = image_input_ids.size(-1)
        ids = torch.arange(past_length, n + past_length, dtype=torch.long, device=self.device)
        return self.image_row_embeddings((ids // self.image_tokens_per_dim)[None]) + self.image_col_embeddings((ids % self.image_tokens_per_dim)[None])

    def forward(self, text_tokens, seg_tokens, img_tokens):
        # zero padding -> unique per-position ids from the vocabulary tail (transformer.py:350-353)
        text_range = (torch.arange(self.text_length) + (self.text_vocab_size - self.text_length)).to(self.device)
        text_tokens = torch.where(text_tokens == 0, text_range, text_tokens)
        text_pos = self.text_pos_embeddings(torch.arange(text_tokens.shape[1], device=self.device))
        embeddings = torch.cat((self.text_token_embedding(text_tokens) + text_pos,
                                self.seg_token_embedding(seg_tokens) + self.get_seg_pos_embeddings(seg_tokens)), dim=1)
        if img_tokens is not None:
            embeddings = torch.cat((embeddings, self.image_token_embedding(img_tokens) + self.get_image_pos_embeddings(img_tokens)), dim=1)
        # transformer.py:366-370 builds tril-with-bidirectional-prefix, which the layers multiply by tril again: causal
        transformer_output, _ = self.transformer(embeddings, None, cache=None, use_cache=False)
        logits = self.to_logits(transformer_output)
        return logits[:, -self.image_length - 1:-1, :]
