"""VQGAN image loss (reference losses/loss_img.py:11-141): L1 (+ perceptual) reconstruction, PatchGAN generator / hinge
discriminator terms, and the adaptive generator weight -- the ratio of two gradient norms at the decoder's last layer, each
obtained with ``torch.autograd.grad(..., retain_graph=True)`` THROUGH the HIP autograd nodes of the decoder (:56-66).

The discriminator runs on the MI355X kernels (``losses.discriminator``).  The two terms of the reference that need pretrained
networks at absolute paths -- LPIPS-VGG16 (``/home/ubuntu/.../vgg.pth``, lpips.py:15) and the face-embedding loss
(face_loss.py:76) -- are pluggable here.  ``perceptual_loss="lpips"`` (the DEFAULT, as the reference's constructor always builds
it, loss_img.py:45) is ``losses.lpips_with_object.LPIPSWithObject`` on the HIP convolutions (weights from ``MAS_LPIPS_CKPT`` /
``MAS_VGG16_CKPT``; missing weights are reported loudly by ``LPIPS.load_from_pretrained``); a callable is used as given; an
explicit ``None`` opts out.  ``face_loss``: the reference always builds ``FaceLoss()`` (loss_img.py:48) -- a pretrained
face-embedding network that is not part of this repository (out of scope) -- so the default here
(``"reference"``) logs ONCE that the face term is absent from the objective and contributes 0; pass a callable to supply it, or
``None`` to opt out silently.  The arithmetic below is the reference's line for line.  ``forward`` keeps the reference's signature and return shapes
(optimizer_idx 0 -> ``loss, (nll_loss, object_loss, face_loss)``; 1 -> ``d_loss``)."""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .discriminator import Discriminator, weights_init


def adopt_weight(weight, global_step, threshold=0, value=0.0):
    if global_step < threshold:
        weight = value
    return weight


def hinge_d_loss(logits_real, logits_fake):
    loss_real = torch.mean(F.relu(1.0 - logits_real))
    loss_fake = torch.mean(F.relu(1.0 + logits_fake))
    return 0.5 * (loss_real + loss_fake)


def vanilla_d_loss(logits_real, logits_fake):
    return 0.5 * (torch.mean(F.softplus(-logits_real)) + torch.mean(F.softplus(logits_fake)))


class VQLPIPSWithDiscriminator(nn.Module):
    _face_warned = False

    def __init__(self, disc_start, codebook_weight=1.0, pixelloss_weight=1.0, disc_factor=1.0, disc_weight=1.0, perceptual_weight=1.0,
                 perceptual_loss="lpips", face_loss="reference"):
        super().__init__()
        self.codebook_weight = codebook_weight
        self.pixel_weight = pixelloss_weight
        if perceptual_loss == "lpips":                 # the reference's default (loss_img.py:45)
            from .lpips_with_object import LPIPSWithObject
            perceptual_loss = LPIPSWithObject().eval()
        self.perceptual_loss = perceptual_loss        # callable(images, reconstructions, bbox_obj) -> per-sample map, or None
        self.perceptual_weight = perceptual_weight
        if isinstance(face_loss, str):
            if face_loss != "reference":
                raise ValueError("face_loss: a callable, None, or 'reference' (the default)")
            if not VQLPIPSWithDiscriminator._face_warned:
                VQLPIPSWithDiscriminator._face_warned = True
                warnings.warn("VQLPIPSWithDiscriminator: the reference adds FaceLoss() (losses/face_loss.py: a pretrained face-embedding "
                              "network, not part of this repository) to the generator objective; it is ABSENT here and contributes 0 "
                              "-- pass face_loss=<callable(images, reconstructions, bbox_face)> to supply it, face_loss=None to "
                              "silence this message")
            face_loss = None
        self.face_loss = face_loss                    # callable(images, reconstructions, bbox_face) -> scalar, or None
        self.discriminator = Discriminator().apply(weights_init)
        self.discriminator_iter_start = disc_start
        self.disc_factor = disc_factor
        self.discriminator_weight = disc_weight

    def calculate_adaptive_weight(self, nll_loss, g_loss, last_layer):
        nll_grads = torch.autograd.grad(nll_loss, last_layer.weight, retain_graph=True)[0]
        g_grads = torch.autograd.grad(g_loss, last_layer.weight, retain_graph=True)[0]
        d_weight = torch.norm(nll_grads) / (torch.norm(g_grads) + 1e-4)
        d_weight = torch.clamp(d_weight, 0.0, 1e4).detach()
        return d_weight * self.discriminator_weight

    def forward(self, optimizer_idx, global_step, images, reconstructions, codebook_loss=None, bbox_obj=None, bbox_face=None,
                last_layer=None):
        if optimizer_idx == 0:  # vqvae loss
            rec_loss = torch.abs(images.contiguous() - reconstructions.contiguous())
            if self.perceptual_loss is not None:
                rec_loss = rec_loss + self.perceptual_weight * self.perceptual_loss(images.contiguous(), reconstructions.contiguous(), bbox_obj)
            nll_loss = torch.mean(rec_loss)
            face_loss = self.face_loss(images, reconstructions, bbox_face) if self.face_loss is not None else images.new_tensor(0)
            object_loss = images.new_tensor(0)
            logits_fake = self.discriminator(reconstructions.contiguous())
            g_loss = -torch.mean(logits_fake)
            d_weight = self.calculate_adaptive_weight(nll_loss, g_loss, last_layer=last_layer)
            disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
            loss = nll_loss + d_weight * disc_factor * g_loss + self.codebook_weight * codebook_loss.mean() + face_loss + object_loss
            return loss, (nll_loss, object_loss, face_loss)
        if optimizer_idx == 1:  # gan loss
            disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
            logits_real = self.discriminator(images.contiguous().detach())
            logits_fake = self.discriminator(reconstructions.contiguous().detach())
            return disc_factor * hinge_d_loss(logits_real, logits_fake)
