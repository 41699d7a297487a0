"""The VQGAN loss stack behind the reference's surface (reference losses/__init__.py:1-2): same import paths and class names, so
``conf/*.yaml`` (``_target_: losses.loss_img.VQLPIPSWithDiscriminator``, ``losses.VQVAEWithBCELoss``) resolve unchanged.  Everything
is this package's own code (round 6: ``loss_seg`` too); nothing here looks for a reference checkout."""
from .loss_img import VQLPIPSWithDiscriminator
from .loss_seg import BCELossWithQuant, VQVAEWithBCELoss
from .lpips import LPIPS
from .lpips_with_object import LPIPSWithObject
