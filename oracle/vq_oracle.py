"""CPU oracle for the VQ-IMG / VQ-SEG hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a functional fp32 restatement (torch CPU ops over a flat
``state_dict``; no ``nn.Module``) of what the reference computes on the path
``VQBASE.forward`` (reference ``models/vqvae.py:36-39``).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package (``make-a-scene_amd/``) never does.

Parity pin: the reference ships no tests / golden vectors, so
the oracle is pinned against outputs of the reference itself, generated in the
authoring container by ``tests/golden/make_golden.py`` (imports
``/root/reference/models`` unmodified) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those fixtures.

Every function cites the reference lines it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# layer plan: which module sits at encoder.model[i] / decoder.model[i]
# --------------------------------------------------------------------------- #
def encoder_plan(channels: Sequence[int], attn_resolutions: Sequence[int], resolution: int,
                 num_res_blocks: int) -> List[Tuple]:
    """Layer list of ``Encoder.__init__`` (reference models/modules.py:217-239).

    Returns tuples ("conv",) | ("res", cin, cout) | ("attn", c) | ("down", c) |
    ("norm", c) | ("swish",) in ``nn.Sequential`` order.  Attention placement follows the
    bookkeeping integer ``resolution`` (modules.py:226,230), not the real map size.
    """
    plan: List[Tuple] = [("conv",)]
    res = resolution
    for i in range(len(channels) - 1):
        cin, cout = channels[i], channels[i + 1]
        for _ in range(num_res_blocks):
            plan.append(("res", cin, cout))
            cin = cout
            if res in attn_resolutions:
                plan.append(("attn", cin))
        if i < len(channels) - 2:
            plan.append(("down", channels[i + 1]))
            res //= 2
    c = channels[-1]
    plan += [("res", c, c), ("attn", c), ("res", c, c), ("norm", c), ("swish",), ("conv",)]
    return plan


def decoder_plan(channels: Sequence[int], attn_resolutions: Sequence[int], resolution: int,
                 num_res_blocks: int) -> List[Tuple]:
    """Layer list of ``Decoder.__init__`` (reference models/modules.py:338-366)."""
    ch_mult = list(channels[1:])
    nres = len(ch_mult)
    block_in = ch_mult[-1]
    curr = resolution // 2 ** (nres - 1)
    plan: List[Tuple] = [("conv",), ("res", block_in, block_in), ("attn", block_in),
                         ("res", block_in, block_in)]
    for i in reversed(range(nres)):
        block_out = ch_mult[i]
        for _ in range(num_res_blocks + 1):
            plan.append(("res", block_in, block_out))
            block_in = block_out
            if curr in attn_resolutions:
                plan.append(("attn", block_in))
        if i > 0:
            plan.append(("up", block_in))
        curr *= 2
    plan += [("norm", block_in), ("swish",), ("conv",)]
    return plan


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def swish(x: Tensor) -> Tensor:
    """x * sigmoid(x)  (modules.py:35-37, 194-196)."""
    return x * torch.sigmoid(x)


def group_norm(sd: SD, p: str, x: Tensor) -> Tensor:
    """GroupNorm(32, C, eps=1e-6, affine)  (modules.py:40-41)."""
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def conv(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet_block(sd: SD, p: str, x: Tensor, cin: int, cout: int) -> Tensor:
    """modules.py:119-136 (dropout p=0; conv_shortcut is never True in the reference)."""
    h = conv(sd, p + ".conv1", swish(group_norm(sd, p + ".norm1", x)), padding=1)
    h = conv(sd, p + ".conv2", swish(group_norm(sd, p + ".norm2", h)), padding=1)
    if cin != cout:
        x = conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """Single-head spatial self-attention (modules.py:167-191)."""
    hn = group_norm(sd, p + ".norm", x)
    q, k, v = conv(sd, p + ".q", hn), conv(sd, p + ".k", hn), conv(sd, p + ".v", hn)
    b, c, h, w = q.shape
    qt = q.reshape(b, c, h * w).transpose(1, 2)              # b, hw, c
    scores = torch.bmm(qt, k.reshape(b, c, h * w)) * (int(c) ** (-0.5))
    probs = torch.softmax(scores, dim=2)                     # over keys
    out = torch.bmm(v.reshape(b, c, h * w), probs.transpose(1, 2)).reshape(b, c, h, w)
    return x + conv(sd, p + ".proj_out", out)


def downsample(sd: SD, p: str, x: Tensor) -> Tensor:
    """zero pad right/bottom by one, 3x3 stride-2 conv (modules.py:75-78)."""
    return conv(sd, p + ".conv", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)


def upsample(sd: SD, p: str, x: Tensor) -> Tensor:
    """nearest x2 then 3x3 conv (modules.py:55-58)."""
    return conv(sd, p + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"), padding=1)


def run_plan(sd: SD, prefix: str, plan: List[Tuple], x: Tensor, taps: Optional[dict] = None) -> Tensor:
    for i, item in enumerate(plan):
        p = f"{prefix}.model.{i}"
        kind = item[0]
        if kind == "conv":
            x = conv(sd, p, x, padding=1)
        elif kind == "res":
            x = resnet_block(sd, p, x, item[1], item[2])
        elif kind == "attn":
            x = attn_block(sd, p, x)
        elif kind == "down":
            x = downsample(sd, p, x)
        elif kind == "up":
            x = upsample(sd, p, x)
        elif kind == "norm":
            x = group_norm(sd, p, x)
        elif kind == "swish":
            x = swish(x)
        else:  # pragma: no cover
            raise ValueError(kind)
        if taps is not None:
            taps[p] = x
    return x


# --------------------------------------------------------------------------- #
# vector quantiser
# --------------------------------------------------------------------------- #
def vq_distances(z_flat: Tensor, codebook: Tensor) -> Tensor:
    """d = sum(z^2) + sum(e^2) - 2 z e^T, fp32, materialised  (modules.py:501-503)."""
    return (torch.sum(z_flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
            - 2 * (z_flat @ codebook.t()))


def codebook_forward(codebook: Tensor, z: Tensor, beta: float = 0.25):
    """Steady-state ``Codebook.forward`` (modules.py:470-517) with the warm-up /
    k-means branches (modules.py:474-499) not taken (``q_counter >= q_re_end``).

    z: [B, C, H, W] fp32.  Returns (z_q [B,C,H,W] with straight-through gradient,
    loss scalar, indices int64 [B*H*W])."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    z_flat = zp.view(-1, codebook.shape[1])
    idx = torch.argmin(vq_distances(z_flat, codebook), dim=1)     # first minimum on ties
    z_q = F.embedding(idx, codebook).view(zp.shape)
    loss = torch.mean((z_q.detach() - zp) ** 2) + beta * torch.mean((z_q - zp.detach()) ** 2)
    z_q = zp + (z_q - zp).detach()
    return z_q.permute(0, 3, 1, 2).contiguous(), loss, idx


def batch_norm_train(sd: SD, p: str, x: Tensor, training: bool = True, eps: float = 1e-5) -> Tensor:
    """``nn.SyncBatchNorm`` with no process group == batch norm (vqvae.py:16).
    Running statistics are not updated here (the oracle is stateless)."""
    if training:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, eps)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, eps)


# --------------------------------------------------------------------------- #
# VQBASE
# --------------------------------------------------------------------------- #
def vqbase_forward(sd: SD, x: Tensor, ddconfig: dict, training: bool = True,
                   quantize: bool = True, taps: Optional[dict] = None):
    """``VQBASE.forward`` (vqvae.py:36-39): encode (vqvae.py:20-24) then decode (:26-29).

    ``quantize=False`` reproduces the warm-up bypass (modules.py:482-484: z returned
    unquantised, loss 0, indices None).  Returns (dec, q_loss, indices, z)."""
    channels = ddconfig.get("channels", [128, 128, 128, 256, 512, 512])
    attn_res = ddconfig.get("attn_resolutions", [32])
    resolution = ddconfig.get("resolution", 512)
    nrb = ddconfig.get("num_res_blocks", 2)
    h = run_plan(sd, "encoder", encoder_plan(channels, attn_res, resolution, nrb), x, taps)
    h = conv(sd, "quant_conv.0", h)
    z = batch_norm_train(sd, "quant_conv.1", h, training)
    if quantize:
        z_q, q_loss, idx = codebook_forward(sd["quantize.embedding.weight"], z)
    else:
        z_q, q_loss, idx = z, z.new_tensor(0), None
    if taps is not None:
        taps["z"], taps["z_q"] = z, z_q
    d = conv(sd, "post_quant_conv", z_q)
    dec = run_plan(sd, "decoder", decoder_plan(channels, attn_res, resolution, nrb), d, taps)
    return dec, q_loss, idx, z


def recon_vq_loss(x: Tensor, dec: Tensor, q_loss: Tensor) -> Tensor:
    """The benchmark / parity loss: L1 term of the VQGAN loss (losses/loss_img.py:79)
    + codebook_weight(=1) * q_loss (losses/loss_img.py:124)."""
    return (x - dec).abs().mean() + q_loss


# --------------------------------------------------------------------------- #
# deterministic, torch-version-independent parameter synthesis
# --------------------------------------------------------------------------- #
def synth_state_dict(ddconfig: dict, n_embed: int, embed_dim: int, seed: int = 0,
                     codebook_scale: Optional[float] = 1.0) -> SD:
    """Builds a VQBASE ``state_dict`` (same keys / shapes as the reference's, SURVEY
    section 8(b)) from a numpy ``RandomState`` so fixtures do not depend on torch's RNG.
    Conv weights ~ U(+-1/sqrt(fan_in)) like torch's default; norm weights ~ 1 + 0.1 N(0,1)
    so the affine terms are exercised.  ``codebook_scale=None`` keeps the reference's
    U(+-1/n_embed) codebook init (modules.py:463)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    sd: SD = {}

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))

    def add_conv(p, cout, cin, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        sd[p + ".weight"] = t(rs.uniform(-bound, bound, size=(cout, cin, k, k)))
        sd[p + ".bias"] = t(rs.uniform(-bound, bound, size=(cout,)))

    def add_norm(p, c):
        sd[p + ".weight"] = t(1.0 + 0.1 * rs.randn(c))
        sd[p + ".bias"] = t(0.1 * rs.randn(c))

    def add_plan(prefix, plan, first_in, first_out, last_in, last_out):
        convs = [i for i, it in enumerate(plan) if it[0] == "conv"]
        for i, it in enumerate(plan):
            p = f"{prefix}.model.{i}"
            if it[0] == "conv":
                if i == convs[0]:
                    add_conv(p, first_out, first_in, 3)
                else:
                    add_conv(p, last_out, last_in, 3)
            elif it[0] == "res":
                add_norm(p + ".norm1", it[1]); add_conv(p + ".conv1", it[2], it[1], 3)
                add_norm(p + ".norm2", it[2]); add_conv(p + ".conv2", it[2], it[2], 3)
                if it[1] != it[2]:
                    add_conv(p + ".nin_shortcut", it[2], it[1], 1)
            elif it[0] == "attn":
                add_norm(p + ".norm", it[1])
                for n in ("q", "k", "v", "proj_out"):
                    add_conv(f"{p}.{n}", it[1], it[1], 1)
            elif it[0] in ("down", "up"):
                add_conv(p + ".conv", it[1], it[1], 3)
            elif it[0] == "norm":
                add_norm(p, it[1])

    channels = ddconfig.get("channels", [128, 128, 128, 256, 512, 512])
    attn_res = ddconfig.get("attn_resolutions", [32])
    resolution = ddconfig.get("resolution", 512)
    nrb = ddconfig.get("num_res_blocks", 2)
    zc = ddconfig.get("z_channels", 256)
    add_plan("encoder", encoder_plan(channels, attn_res, resolution, nrb),
             ddconfig.get("in_channels", 3), channels[0], channels[-1], zc)
    dplan = decoder_plan(channels, attn_res, resolution, nrb)
    add_plan("decoder", dplan, zc, channels[-1], dplan[-3][1], ddconfig.get("out_channels", 3))
    if codebook_scale is None:
        sd["quantize.embedding.weight"] = t(rs.uniform(-1.0 / n_embed, 1.0 / n_embed, size=(n_embed, embed_dim)))
    else:
        sd["quantize.embedding.weight"] = t(codebook_scale * rs.randn(n_embed, embed_dim))
    add_conv("quant_conv.0", embed_dim, zc, 1)
    add_norm("quant_conv.1", embed_dim)
    sd["quant_conv.1.running_mean"] = torch.zeros(embed_dim)
    sd["quant_conv.1.running_var"] = torch.ones(embed_dim)
    sd["quant_conv.1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    add_conv("post_quant_conv", zc, embed_dim, 1)
    return sd


def synth_image_batch(batch: int, channels: int, size: int, seed: int = 0) -> Tensor:
    """Seeded synthetic batch in [0,1) (``torch.rand`` stand-in made
    torch-RNG independent)."""
    import numpy as np
    rs = np.random.RandomState(1000 + seed)
    return torch.from_numpy(rs.rand(batch, channels, size, size).astype(np.float32))
