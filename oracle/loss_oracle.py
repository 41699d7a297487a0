"""CPU oracle for the VQGAN loss stack  --  TEST INFRASTRUCTURE ONLY.

Functional fp32 restatement (torch CPU ops over a flat ``state_dict``) of
  * ``Discriminator.forward``  -- reference losses/discriminator.py:17-38: Conv(3->64,k4,s2,p1) LeakyReLU(0.2), three
    [Conv(k4, s2/s2/s1, p1, no bias) BatchNorm2d LeakyReLU(0.2)] blocks (64->128->256->512), Conv(512->1,k4,s1,p1);
  * ``hinge_d_loss`` / ``vanilla_d_loss`` / ``adopt_weight`` -- losses/loss_img.py:11-31;
  * ``calculate_adaptive_weight`` -- losses/loss_img.py:56-66;
  * the generator / discriminator branches of ``VQLPIPSWithDiscriminator.forward`` -- :68-141, without the two terms that need
    pretrained networks at absolute paths (LPIPS lpips.py:15, face loss face_loss.py:76): perceptual_weight * p_loss and
    face_loss are taken as 0, object_loss is 0 in the reference itself (:90).
Pinned against the reference's own Discriminator / loss functions in tests/golden/disc_tiny.npz (make_loss_golden.py imports
losses/discriminator.py by file path and restates nothing).  Only tests may import this module."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def disc_layout(in_channels=3, nf=64, n_layers=3):
    """(index in nn.Sequential, kind, cin, cout, stride, has_bias) per layer -- discriminator.py:20-36"""
    out = [(0, "conv", in_channels, nf, 2, True), (1, "lrelu")]
    mult, idx = 1, 2
    for i in range(1, n_layers + 1):
        last, mult = mult, min(2 ** i, 8)
        out += [(idx, "conv", nf * last, nf * mult, 2 if i < n_layers else 1, False), (idx + 1, "bn", nf * mult), (idx + 2, "lrelu")]
        idx += 3
    out.append((idx, "conv", nf * mult, 1, 1, True))
    return out


def synth_disc_state_dict(seed=0, in_channels=3, nf=64, n_layers=3) -> SD:
    """weights_init-like synthetic parameters from numpy RandomState (independent of torch's RNG stream); BatchNorm running
    statistics non-trivial so eval mode is exercised too"""
    rs = np.random.RandomState(seed)
    sd: SD = {}
    for lay in disc_layout(in_channels, nf, n_layers):
        i, kind = lay[0], lay[1]
        if kind == "conv":
            _, _, cin, cout, _, has_bias = lay
            sd[f"model.{i}.weight"] = torch.from_numpy((0.02 * rs.randn(cout, cin, 4, 4) * (3.0 if cin > 3 else 8.0)).astype(np.float32))
            if has_bias:
                sd[f"model.{i}.bias"] = torch.from_numpy((0.05 * rs.randn(cout)).astype(np.float32))
        elif kind == "bn":
            c = lay[2]
            sd[f"model.{i}.weight"] = torch.from_numpy((1.0 + 0.1 * rs.randn(c)).astype(np.float32))
            sd[f"model.{i}.bias"] = torch.from_numpy((0.05 * rs.randn(c)).astype(np.float32))
            sd[f"model.{i}.running_mean"] = torch.from_numpy((0.1 * rs.randn(c)).astype(np.float32))
            sd[f"model.{i}.running_var"] = torch.from_numpy((1.0 + 0.2 * rs.rand(c)).astype(np.float32))
            sd[f"model.{i}.num_batches_tracked"] = torch.tensor(3, dtype=torch.int64)
    return sd


class _RoundBF16(torch.autograd.Function):
    """bf16 storage of a tensor AND of its gradient (what the MI355X path does between kernels)"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def disc_forward(sd: SD, x: Tensor, training: bool = True, in_channels=3, nf=64, n_layers=3, bf16_storage: bool = False) -> Tensor:
    """``bf16_storage``: the same arithmetic with every inter-layer tensor (and its gradient) and every conv weight rounded to
    bf16, fp32 accumulation, fp32 logits -- the precision model of the MI355X path, used to separate "bf16 storage error" (large
    for gradients that cross three BatchNorms: 13 % rel-L2 on the input gradient of the golden case) from kernel error."""
    r = _RoundBF16.apply if bf16_storage else (lambda t: t)
    h = x
    last = disc_layout(in_channels, nf, n_layers)[-1][0]
    for lay in disc_layout(in_channels, nf, n_layers):
        i, kind = lay[0], lay[1]
        if kind == "conv":
            w = sd[f"model.{i}.weight"]
            h = F.conv2d(r(h), r(w) if bf16_storage else w, sd.get(f"model.{i}.bias"), stride=lay[4], padding=1)
            if i != last:
                h = r(h)
        elif kind == "bn":       # nn.BatchNorm2d, eps 1e-5; training: batch statistics (running buffers are not touched here)
            # (bf16_storage: no rounding here -- since round 6 the MI355X path normalises and activates in ONE pass, fp32 inside, and
            #  rounds the activated map once: the rounding sits behind the LeakyReLU below)
            h = F.batch_norm(h, None if training else sd[f"model.{i}.running_mean"], None if training else sd[f"model.{i}.running_var"],
                             sd[f"model.{i}.weight"], sd[f"model.{i}.bias"], training, 0.1, 1e-5)
        else:
            h = r(F.leaky_relu(h, 0.2))
    return h


def adopt_weight(weight, global_step, threshold=0, value=0.0):            # loss_img.py:11-14
    return value if global_step < threshold else weight


def hinge_d_loss(logits_real: Tensor, logits_fake: Tensor) -> Tensor:     # loss_img.py:17-21
    return 0.5 * (torch.mean(F.relu(1.0 - logits_real)) + torch.mean(F.relu(1.0 + logits_fake)))


def vanilla_d_loss(logits_real: Tensor, logits_fake: Tensor) -> Tensor:   # loss_img.py:24-29
    return 0.5 * (torch.mean(F.softplus(-logits_real)) + torch.mean(F.softplus(logits_fake)))


def adaptive_weight(nll_loss: Tensor, g_loss: Tensor, last_weight: Tensor, disc_weight: float) -> Tensor:   # loss_img.py:56-66
    nll_grads = torch.autograd.grad(nll_loss, last_weight, retain_graph=True)[0]
    g_grads = torch.autograd.grad(g_loss, last_weight, retain_graph=True)[0]
    d_weight = torch.norm(nll_grads) / (torch.norm(g_grads) + 1e-4)
    return torch.clamp(d_weight, 0.0, 1e4).detach() * disc_weight


def generator_loss(sd_disc: SD, images: Tensor, rec: Tensor, codebook_loss: Tensor, last_weight: Tensor, global_step: int, disc_start: int,
                   codebook_weight=1.0, disc_factor=1.0, disc_weight=1.0):
    """optimizer_idx == 0 (loss_img.py:79-129) with p_loss = face_loss = 0; returns (loss, nll_loss, g_loss, d_weight)"""
    nll = torch.mean(torch.abs(images - rec))
    g_loss = -torch.mean(disc_forward(sd_disc, rec, True))
    d_w = adaptive_weight(nll, g_loss, last_weight, disc_weight)
    loss = nll + d_w * adopt_weight(disc_factor, global_step, disc_start) * g_loss + codebook_weight * codebook_loss.mean()
    return loss, nll, g_loss, d_w


def discriminator_loss(sd_disc: SD, images: Tensor, rec: Tensor, global_step: int, disc_start: int, disc_factor=1.0) -> Tensor:
    """optimizer_idx == 1 (loss_img.py:132-141)"""
    return adopt_weight(disc_factor, global_step, disc_start) * hinge_d_loss(disc_forward(sd_disc, images.detach(), True),
                                                                           disc_forward(sd_disc, rec.detach(), True))
