"""The VQ-SEG stage's reconstruction losses behind the reference's names (reference losses/loss_seg.py:6-41; ``conf/seg_config.yaml``'s
``_target_: losses.VQVAEWithBCELoss``).  Elementwise torch, off the hot path: restated here so that this package never needs a
reference checkout at run time (until round 5 ``losses/__init__.py`` fell through to the reference's own file).  Pinned against the
reference: ``tests/golden/loss_seg.npz`` (``tests/golden/make_golden_r6.py`` runs the reference's classes), ``tests/test_losses_host.py``.

Both classes weigh the positive class of the five channels 153..157 twenty-fold (``pos_weight`` of the logits BCE; a persistent buffer
named ``weight``, so ``state_dict`` carries it as the reference's does) and add ``codebook_weight * qloss``;
``VQVAEWithBCELoss`` adds the mean squared error of the sigmoid as well."""
import torch
import torch.nn.functional as F
from torch import nn

_HEAVY_CHANNELS = (153, 158)        # half-open channel range with positive weight 20
_HEAVY_WEIGHT = 20.0


class _SegLossBase(nn.Module):
    def __init__(self, image_channels=159, codebook_weight=1.0):
        super().__init__()
        self.codebook_weight = codebook_weight
        w = torch.ones(image_channels)
        w[_HEAVY_CHANNELS[0]:_HEAVY_CHANNELS[1]] = _HEAVY_WEIGHT
        self.register_buffer("weight", w)

    def _bce(self, target, prediction):
        # channels last, so that the per-channel pos_weight broadcasts over (N, H, W)
        return F.binary_cross_entropy_with_logits(prediction.movedim(1, -1), target.movedim(1, -1), pos_weight=self.weight)


class BCELossWithQuant(_SegLossBase):
    def forward(self, qloss, target, prediction):
        return self._bce(target, prediction) + self.codebook_weight * qloss


class VQVAEWithBCELoss(_SegLossBase):
    def forward(self, qloss, target, prediction):
        rec = F.mse_loss(torch.sigmoid(prediction), target) + self._bce(target, prediction)
        return rec + self.codebook_weight * qloss
