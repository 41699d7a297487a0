"""The autoregressive transformer behind the reference's class surface (reference models/transformer.py):
``SelfAttention`` :17, ``MLP`` :118, ``TransformerLayer`` :142, ``Transformer`` :213, ``MakeAScene`` :275 --
same constructor arguments, submodule names and ``state_dict`` keys (incl. the ``transformer.mask`` buffer).

Training configuration (pb-relax on, rudalle-relax off, sandwich layer-norm on -- the reference's defaults): the
attention core is the flash-style HIP kernel (``mas_attn_causal_fwd``); the PB-relax shift
(:64-70) subtracts a per-(batch, head) constant before the softmax and the per-layer ``mask * tril`` product (:260-263)
is pure causal, so an online-softmax causal kernel computes the same function.  Linear / LayerNorm /
embeddings are plain library calls (hipBLASLt GEMMs), as SURVEY 2.1 K11 allows.

Sampling configuration (reference transformer.py:73-115,170-210 ``use_cache`` / ``cache``): KV-cached
token-by-token decoding on ``mas_attn_decode`` (HBM-bound, one pass over the cached keys / values per new token).  The
reference's cached branch cannot run as committed (``TransformerLayer`` passes ``cache`` / ``use_cache`` to
``SelfAttention.forward`` in swapped positions, :181, and feeds the MLP the CACHED rows instead of the new ones, :199-200;
SURVEY Appendix B) and nothing in the reference ever calls it with ``use_cache=True``; what is kept is its calling
convention -- the layer receives the FULL sequence and a per-layer cache tuple whose first entry has the cached length on
axis -2, computes only the new positions and returns full-length outputs -- and what is pinned is the only behaviour a
cache can have: the logits of cached decoding equal those of the uncached forward (tests/test_gpu_sampling.py, and through
it the reference's golden logits)."""
import math
import os

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

    def forward(self, x, residual=None, producer_bias_grad=False):
        """``producer_bias_grad``: x is the output of a ``Linear`` (out_proj / lin2 in front of the sandwich LayerNorms): the backward
        kernel then also produces that layer's bias gradient (the column sums of its dx) instead of a separate pass over dx."""
        if not self.elementwise_affine or len(self.normalized_shape) != 1:
            raise NotImplementedError("libmas_hip LayerNorm: affine, over the last dimension")
        return ops.layer_norm(x, self.weight, self.bias, self.eps, residual, producer_bias_grad=producer_bias_grad)

    def fork(self, x):
        """-> (LN(x), x): the normalised tensor and the skip connection as outputs of ONE autograd node (their gradients are added
        inside the LayerNorm backward kernel)."""
        if not self.elementwise_affine or len(self.normalized_shape) != 1:
            raise NotImplementedError("libmas_hip LayerNorm: affine, over the last dimension")
        return ops.layer_norm_fork(x, self.weight, self.bias, self.eps)


class Linear(nn.Linear):
    """nn.Linear parameters / state_dict keys.  Under bf16 autocast on the GPU (the mode the transformer is trained and benched
    in) the layer is one autograd node around the library GEMMs with a fp32 weight gradient and the HIP column-sum bias gradient
    (``ops.linear_bf16``); in every other mode it is nn.Linear."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.bias is not None:
            ops._bf16_shadows.register(self.weight)
            ops._bf16_shadows.register(self.bias)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        ops.drop_weight_cache_of(self.weight, self.bias)

    def forward(self, x):
        if (x.is_cuda and self.bias is not None and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.weight.dtype == torch.float32 and self.in_features % 8 == 0 and self.out_features % 8 == 0):
            return ops.linear_bf16(x, self.weight, self.bias)
        return super().forward(x)


def _kv_backing(t, b, d):
    """[B, cap, D] buffer behind a cached k / v entry [B, H, L, hd].  The entries this module returns are views of such a buffer
    (``t._base``); anything else -- a clone, a contiguous copy, an index_select over the batch (beam reorder) -- has lost it and is
    laid out [B, H, L, hd]: the buffer is rebuilt from its contents (capacity = L) instead of being reinterpreted."""
    base = t._base
    if base is not None and base.dim() == 3 and base.shape[0] == b and base.shape[2] == d and base.is_contiguous() \
            and t.data_ptr() == base.data_ptr() and t.stride(2) == d:
        return base
    return t.permute(0, 2, 1, 3).reshape(b, t.shape[2], d).contiguous()


def _row_backing(t, b, d):
    """[B, cap, D] buffer behind a cached [B, L, D] row view (attention outputs / layer outputs so far), or a copy of its rows"""
    base = t._base
    if base is not None and base.dim() == 3 and base.shape[0] == b and base.shape[2] == d and base.is_contiguous() \
            and t.data_ptr() == base.data_ptr():
        return base
    return t.contiguous()


class SelfAttention(nn.Module):
    def __init__(self, hidden_dim, num_attn_heads, attn_dropout_prob, out_dropout_prob, cogview_pb_relax=True, rudalle_relax=False):
        super(SelfAttention, self).__init__()
        self.hidden_dim = hidden_dim
        self.num_attn_heads = num_attn_heads
        self.d = math.sqrt(self.hidden_dim // self.num_attn_heads)
        self.qkv = Linear(hidden_dim, 3 * hidden_dim)
        self.attn_drop = nn.Dropout(attn_dropout_prob)
        self.out_proj = Linear(hidden_dim, hidden_dim)
        self.out_drop = nn.Dropout(out_dropout_prob)
        self.cogview_pb_relax = cogview_pb_relax
        self.rudalle_relax = rudalle_relax

    def forward(self, x, mask, use_cache=False, cache=None):
        if self.rudalle_relax or (self.training and self.attn_drop.p > 0):
            raise NotImplementedError("rudalle_relax / attention dropout are off the measured path (reference defaults: off)")
        if not use_cache:
            qkv = self.qkv(x)
            context = ops.causal_attention(qkv, self.num_attn_heads)     # `mask` is causal by construction (transformer.py:260-263)
            return self.out_drop(self.out_proj(context)), cache
        return self._forward_cached(x, cache)

    # ---- KV-cached inference (reference transformer.py:74-84,106-111) ------------------------------------------------------
    # cache = (k, v, out): k, v are [B, H, L, hd] VIEWS (the reference's shapes: cached length on axis -2) of preallocated
    # [B, S_max, H*hd] buffers in the layout nn.Linear emits, `out` is a [B, L, D] view of the projected outputs so far.
    # x is the FULL sequence (reference convention); only x[:, L:] is computed.
    def _forward_cached(self, x, cache):
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError("KV-cached attention is an inference path: call it under torch.no_grad()")
        b, s_tot, d = x.shape
        h = self.num_attn_heads
        past = 0 if cache is None else cache[0].shape[-2]
        nq = s_tot - past
        if nq <= 0:
            raise RuntimeError(f"SelfAttention: the cache already holds {past} positions, the input has {s_tot}")
        qkv = self.qkv(x[:, past:, :])
        dt = qkv.dtype
        if cache is None:
            cap = max(getattr(self, "cache_capacity", 0), s_tot)
            kbuf = torch.empty((b, cap, d), dtype=dt, device=x.device)
            vbuf = torch.empty_like(kbuf)
            obuf = torch.empty_like(kbuf)
        else:
            # the [B, cap, D] buffers behind the views; an entry that is NOT such a view any more (cloned, made contiguous,
            # index_select'ed for a beam / batch reorder) is rebuilt from its [B, H, L, hd] / [B, L, D] contents
            kbuf, vbuf = (_kv_backing(t, b, d) for t in cache[:2])
            obuf = _row_backing(cache[2], b, d)
            if kbuf.shape[1] < s_tot or obuf.shape[1] < s_tot or vbuf.shape[1] != kbuf.shape[1]:                                                  # grow geometrically, copy once
                cap = max(2 * kbuf.shape[1], s_tot)
                grow = lambda t: torch.cat([t[:, :past], torch.empty((b, cap - past, d), dtype=t.dtype, device=t.device)], dim=1)
                kbuf, vbuf, obuf = grow(kbuf), grow(vbuf), grow(obuf)
        kbuf[:, past:s_tot] = qkv[..., d:2 * d]
        vbuf[:, past:s_tot] = qkv[..., 2 * d:]
        q = qkv[..., :d]
        if cache is None and nq > 1:
            context = ops.causal_attention(qkv, h)                  # prefill: the training kernel on the whole prompt
        else:
            context = ops.attention_decode(q, kbuf, vbuf, past, h)  # decode: one pass over the cached rows
        obuf[:, past:s_tot] = self.out_proj(context)
        hd = d // h
        view = lambda t: t[:, :s_tot].view(b, s_tot, h, hd).permute(0, 2, 1, 3)
        return self.out_drop(obuf[:, :s_tot]), (view(kbuf), view(vbuf), obuf[:, :s_tot])


class MLP(nn.Module):
    def __init__(self, hidden_dim, dropout_prob, rudalle_relax=False):
        super(MLP, self).__init__()
        self.lin1 = Linear(hidden_dim, 4 * hidden_dim)
        self.lin2 = Linear(4 * hidden_dim, hidden_dim)
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

    def forward(self, x, mask, cache=None, use_cache=False, mlp_cache=False, pre=None, next_ln=None):
        """``pre`` / ``next_ln`` (round 6, training / uncached path): the pre-LayerNorm of a sub-block rides in the launch of the sandwich
        LayerNorm + residual that produces its input (``ops.layer_norm_pair``).  ``pre`` = (ln_in(x), x) already computed by the previous
        layer's last launch; ``next_ln`` = the module to apply to this layer's output (the next layer's ``ln_in`` or the final LayerNorm):
        the layer then returns a third value, (next_ln(y), y).  Without them the layer behaves as before (two values returned)."""
        if use_cache:
            return self._forward_cached(x, cache)
        fork = not self.cogview_layernorm_prescale      # (the prescale variant divides x before the LayerNorm: plain path)
        pair = fork and self.cogview_sandwich_layernorm
        if pre is not None:
            ln, skip = pre
        else:
            ln, skip = self.ln_in.fork(x) if fork else (self.ln_in(self._prescale(x)), x)
        attn_out, new_cache = self.attn(ln, mask, False, cache)
        if pair:                                        # x + LN(attn_out) and ln_out of the sum: one pass
            ln, skip = ops.layer_norm_pair(attn_out, skip, self.first_ln_sandwich, self.ln_out, producer_bias_grad=True)
        else:
            if self.cogview_sandwich_layernorm:
                x = self.first_ln_sandwich(self._prescale(attn_out), residual=skip,     # x + LN(attn_out), one pass
                                           producer_bias_grad=not self.cogview_layernorm_prescale)
            else:
                x = skip + attn_out
            ln, skip = self.ln_out.fork(x) if fork else (self.ln_out(self._prescale(x)), x)
        mlp_out = self.mlp(ln)
        if pair and next_ln is not None:
            ln_next, y = ops.layer_norm_pair(mlp_out, skip, self.second_ln_sandwich, next_ln, producer_bias_grad=True)
            return y, new_cache, (ln_next, y)
        if self.cogview_sandwich_layernorm:
            y = self.second_ln_sandwich(mlp_out, residual=skip, producer_bias_grad=not self.cogview_layernorm_prescale)
        else:
            y = skip + mlp_out
        return (y, new_cache) if next_ln is None else (y, new_cache, None)

    def _forward_cached(self, x, cache):
        """Full sequence in, full sequence out, only the positions past the cache computed (reference transformer.py:170-210
        with its two defects fixed, see the module docstring).  cache = (k, v, attn_out, layer_out): the attention's tuple plus
        this layer's own outputs so far (the reference keeps those for the last layer only, ``mlp_cache``)."""
        past = 0 if cache is None else cache[0].shape[-2]
        x_new = x[:, past:, :]
        ln = self.ln_in(self._prescale(x_new))
        if cache is not None:      # the attention slices its input at the cached length: hand it a full-length tensor whose
            ln_full = x.new_empty(x.shape[:-1] + (ln.shape[-1],), dtype=ln.dtype)   # cached rows are never read
            ln_full[:, past:] = ln
        else:
            ln_full = ln
        attn_full, kv = self.attn(ln_full, None, True, None if cache is None else cache[:3])
        attn_new = attn_full[:, past:, :]
        if self.cogview_sandwich_layernorm:
            h = self.first_ln_sandwich(self._prescale(attn_new), residual=x_new)
        else:
            h = x_new + attn_new
        mlp_out = self.mlp(self.ln_out(self._prescale(h)))
        y_new = self.second_ln_sandwich(mlp_out, residual=h) if self.cogview_sandwich_layernorm else h + mlp_out
        if cache is None:
            ybuf = torch.empty((x.shape[0], _kv_backing(kv[0], x.shape[0], x.shape[-1]).shape[1], x.shape[-1]), dtype=y_new.dtype,
                               device=x.device)
        else:
            ybuf = _row_backing(cache[3], x.shape[0], x.shape[-1])
            if ybuf.shape[1] < x.shape[1]:                        # same growth rule as the attention's buffers
                cap = max(2 * ybuf.shape[1], x.shape[1])
                ybuf = torch.cat([ybuf[:, :past], torch.empty((x.shape[0], cap - past, x.shape[-1]), dtype=ybuf.dtype,
                                                              device=x.device)], dim=1)
        ybuf[:, past:x.shape[1]] = y_new
        return ybuf[:, :x.shape[1]], kv + (ybuf[:, :x.shape[1]],)


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
        past = 0
        if use_cache and cache.get(0) is not None:
            past = cache[0][0].shape[-2]
        if not use_cache:
            # training / uncached forward: every pre-LayerNorm (each layer's ln_in, and the final LayerNorm) is computed by the launch
            # that produced its input -- the previous layer's second sandwich LayerNorm + residual (TransformerLayer.forward)
            pre = None
            n = len(self.layers)
            for i, layer in enumerate(self.layers):
                nxt = self.layers[i + 1].ln_in if i + 1 < n else self.final_ln
                x, layer_cache, pre = layer(x, None, cache.get(i), mlp_cache=i == n - 1, use_cache=use_cache, pre=pre, next_ln=nxt)
                cache[i] = layer_cache
            return (pre[0] if pre is not None else self.final_ln(x)), cache
        for i, layer in enumerate(self.layers):
            # attn_mask * self.mask is pure causal: nothing to materialise for the kernel
            x, layer_cache = layer(x, None, cache.get(i), mlp_cache=i == len(self.layers) - 1, use_cache=use_cache)
            cache[i] = layer_cache
        # cached decoding: the final LayerNorm of the positions computed by THIS call only (everything the caller can need:
        # rows are independent, and the earlier ones were returned by the earlier calls)
        return self.final_ln(x[:, past:] if past else x), cache


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
        self.to_logits = torch.nn.Sequential(LayerNorm(hidden_dim), Linear(hidden_dim, image_vocab_size))

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

    def _prompt_embeddings(self, text_tokens, seg_tokens):
        # zero padding -> unique per-position ids from the vocabulary tail (transformer.py:350-353)
        text_range = (torch.arange(self.text_length) + (self.text_vocab_size - self.text_length)).to(self.device)
        text_tokens = torch.where(text_tokens == 0, text_range, text_tokens)
        text_pos = self.text_pos_embeddings(torch.arange(text_tokens.shape[1], device=self.device))
        return torch.cat((self.text_token_embedding(text_tokens) + text_pos,
                          self.seg_token_embedding(seg_tokens) + self.get_seg_pos_embeddings(seg_tokens)), dim=1)

    @torch.no_grad()
    def generate(self, text_tokens, seg_tokens, temperature=1.0, top_k=None, cond_scale=None, generator=None, img_tokens=None,
                 return_logits=False):
        """Autoregressive sampling of the ``image_length`` image tokens given text + segmentation tokens, KV-cached: one prefill
        over the prompt (the training attention kernel), then one ``mas_attn_decode`` pass per layer and token.
        ``temperature`` 0 -> greedy; ``top_k`` keeps the k most likely tokens; ``cond_scale`` s -> classifier-free guidance
        against the text-free stream the reference trains for (train.py:147-148 zeroes the text with probability ``uncond_p``):
        logits = l_uncond + s * (l_cond - l_uncond).  ``img_tokens`` [B, image_length]: teacher forcing (the given tokens are fed
        instead of the sampled ones -- used by the tests to compare every step's logits with the uncached forward).
        Returns the tokens [B, image_length] (int64) -- ``VQBASE.decode_code`` turns them into an image -- and, with
        ``return_logits``, the logits [B, image_length, vocab] they were drawn from."""
        b = text_tokens.shape[0]
        guided = cond_scale is not None
        if guided:
            text_tokens = torch.cat([text_tokens, torch.zeros_like(text_tokens)], dim=0)
            seg_tokens = torch.cat([seg_tokens, seg_tokens], dim=0)
        prompt = self._prompt_embeddings(text_tokens, seg_tokens)
        bb, plen, d = prompt.shape
        for layer in self.transformer.layers:
            layer.attn.cache_capacity = self.total_length
        buf = prompt.new_empty((bb, self.total_length, d))
        buf[:, :plen] = prompt
        cur = plen
        hidden, cache = self.transformer(buf[:, :cur], None, cache={}, use_cache=True)
        tokens = torch.empty((b, self.image_length), dtype=torch.long, device=prompt.device)
        all_logits = [] if return_logits else None
        for i in range(self.image_length):
            logits = self.to_logits(hidden[:, -1:, :])[:, 0, :].float()
            if guided:
                logits = logits[b:] + float(cond_scale) * (logits[:b] - logits[b:])
            if return_logits:
                all_logits.append(logits)
            if img_tokens is not None:
                tok = img_tokens[:, i]
            elif temperature == 0:
                tok = logits.argmax(dim=-1)
            else:
                lg = logits / float(temperature)
                if top_k is not None:
                    kth = torch.topk(lg, int(top_k), dim=-1).values[:, -1:]
                    lg = lg.masked_fill(lg < kth, float("-inf"))
                tok = torch.multinomial(torch.softmax(lg, dim=-1), 1, generator=generator)[:, 0]
            tokens[:, i] = tok
            if i + 1 == self.image_length:
                break
            t_in = torch.cat([tok, tok], dim=0) if guided else tok
            emb = self.image_token_embedding(t_in[:, None]) + self.get_image_pos_embeddings(t_in[:, None], past_length=i)
            buf[:, cur] = emb[:, 0]
            cur += 1
            hidden, cache = self.transformer(buf[:, :cur], None, cache=cache, use_cache=True)
        if return_logits:
            return tokens, torch.stack(all_logits, dim=1)
        return tokens

    def forward(self, text_tokens, seg_tokens, img_tokens):
        embeddings = self._prompt_embeddings(text_tokens, seg_tokens)
        if img_tokens is not None:
            embeddings = torch.cat((embeddings, self.image_token_embedding(img_tokens) + self.get_image_pos_embeddings(img_tokens)), dim=1)
        # transformer.py:366-370 builds tril-with-bidirectional-prefix, which the layers multiply by tril again: causal
        transformer_output, _ = self.transformer(embeddings, None, cache=None, use_cache=False)
        logits = self.to_logits(transformer_output)
        return logits[:, -self.image_length - 1:-1, :]
