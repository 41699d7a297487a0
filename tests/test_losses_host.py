"""Host-side contract of the loss stack (no GPU): the reference's constructor defaults and what happens to missing weights."""
import os
import warnings

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the `loss:` block of the reference's conf/img_config.yaml:57-62, verbatim
IMG_CONFIG_LOSS_BLOCK = """
loss:
  #_target_: losses.VQVAEWithBCELoss
  _target_: losses.loss_img.VQLPIPSWithDiscriminator
  disc_start: 250001
  disc_weight: 0.8
  codebook_weight: 1.0
"""


def _instantiate(block):
    import importlib
    cfg = dict(yaml.safe_load(block)["loss"])
    mod, cls = cfg.pop("_target_").rsplit(".", 1)
    return getattr(importlib.import_module(mod), cls)(**cfg)


def test_yaml_loss_block_builds_the_reference_objective():
    """reference losses/loss_img.py:44-48 ALWAYS builds LPIPSWithObject and FaceLoss: the drop-in must not silently train
    L1 + GAN only.  The perceptual term is built by default; the face term (out of scope: a pretrained network that is not in
    the repository) is absent and SAYS so."""
    from losses import loss_img
    from losses.lpips import LPIPS
    loss_img.VQLPIPSWithDiscriminator._face_warned = False
    LPIPS._warned = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lf = _instantiate(IMG_CONFIG_LOSS_BLOCK)
    assert lf.perceptual_loss is not None and type(lf.perceptual_loss).__name__ == "LPIPSWithObject"
    assert not lf.perceptual_loss.training                                  # .eval(), loss_img.py:45
    assert lf.discriminator_iter_start == 250001 and lf.discriminator_weight == 0.8 and lf.codebook_weight == 1.0
    msgs = [str(w.message) for w in rec]
    assert any("FaceLoss" in m and "ABSENT" in m for m in msgs)
    assert any("RANDOM initialisation" in m for m in msgs)                  # no checkpoint here: reported, by name
    assert lf.face_loss is None
    # explicit opt-outs are silent
    with warnings.catch_warnings(record=True) as rec2:
        warnings.simplefilter("always")
        l0 = loss_img.VQLPIPSWithDiscriminator(disc_start=0, perceptual_loss=None, face_loss=None)
    assert l0.perceptual_loss is None and not rec2


def test_default_construction_without_weights_is_an_error(monkeypatch):
    """a default run must not optimise a random-feature 'perceptual' term silently -- without MAS_LPIPS_CKPT /
    MAS_VGG16_CKPT the default constructor raises, like the reference's (whose weight files are absolute paths, lpips.py:15)."""
    from losses import loss_img, lpips
    monkeypatch.delenv("MAS_LPIPS_STRICT", raising=False)
    monkeypatch.delenv("MAS_VGG16_CKPT", raising=False)
    monkeypatch.setattr(lpips, "CKPT_PATHS", ())
    with pytest.raises(RuntimeError, match="RANDOM initialisation"):
        loss_img.VQLPIPSWithDiscriminator(disc_start=0, face_loss=None)
    loss_img.VQLPIPSWithDiscriminator(disc_start=0, perceptual_loss=None, face_loss=None)      # the explicit opt-out builds


def test_lpips_partial_checkpoint_is_reported(tmp_path, monkeypatch):
    """the reference's vgg.pth holds the five lin heads only (its backbone comes from torchvision); loading it -- or any
    partial MAS_LPIPS_CKPT -- must not pass for a loaded network.  MAS_VGG16_CKPT supplies the backbone in torchvision's key
    layout; MAS_LPIPS_STRICT=1 raises while anything is missing."""
    from losses import lpips
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = lpips.LPIPS()
    heads = {k: v.clone() + 1.0 for k, v in full.state_dict().items() if k.startswith("lin")}
    hp = tmp_path / "vgg.pth"
    torch.save(heads, hp)
    monkeypatch.setattr(lpips, "CKPT_PATHS", (str(hp),))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m = lpips.LPIPS()
    assert len(m.unloaded) == 26 and all(k.startswith("vgg.") for k in m.unloaded)
    assert any("backbone: 26 of 26" in str(w.message) for w in rec)
    assert torch.equal(m.lin3.model[1].weight, heads["lin3.model.1.weight"])
    monkeypatch.setenv("MAS_LPIPS_STRICT", "1")
    with pytest.raises(RuntimeError, match="RANDOM initialisation"):
        lpips.LPIPS()
    monkeypatch.delenv("MAS_LPIPS_STRICT")
    # the backbone in torchvision's layout: features.<idx>.{weight,bias}
    tv = {}
    for idx, (sl, j) in lpips.LPIPS._TV_CONVS.items():
        for leaf in ("weight", "bias"):
            tv[f"features.{idx}.{leaf}"] = full.state_dict()[f"vgg.{sl}.{j}.{leaf}"].clone() * 0.5
    tp = tmp_path / "vgg16_tv.pth"
    torch.save(tv, tp)
    monkeypatch.setenv("MAS_VGG16_CKPT", str(tp))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m2 = lpips.LPIPS()
    assert m2.unloaded == [] and not rec
    assert torch.equal(m2.vgg.slice3[3].weight, tv["features.12.weight"])


def test_loss_seg_vs_reference_golden():
    """losses.loss_seg (BCELossWithQuant, VQVAEWithBCELoss) against the reference's own classes (reference losses/loss_seg.py:6-41;
    fixture: tests/golden/make_golden_r6.py): loss values, gradients w.r.t. the prediction, the ``weight`` buffer and the state_dict keys."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_r6 import loss_seg_inputs
    import losses
    g = np.load(os.path.join(ROOT, "tests", "golden", "loss_seg.npz"))
    pred, target, qloss = loss_seg_inputs()
    for name in ("BCELossWithQuant", "VQVAEWithBCELoss"):
        for cw in (1.0, 0.25):
            m = getattr(losses, name)(image_channels=159, codebook_weight=cw)
            p = torch.from_numpy(pred).requires_grad_(True)
            loss = m(torch.tensor(qloss), torch.from_numpy(target), p)
            loss.backward()
            assert abs(float(loss) - float(g[f"{name}:{cw}:loss"])) < 1e-6 * max(1.0, abs(float(loss)))
            ref = torch.from_numpy(g[f"{name}:{cw}:grad"])
            assert float((p.grad - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        assert np.array_equal(m.weight.numpy(), g[f"{name}:weight"]) and float(m.weight[153]) == 20.0 and float(m.weight[158]) == 1.0
        assert sorted(m.state_dict().keys()) == list(g[f"{name}:state_keys"])
    from losses.loss_seg import VQVAEWithBCELoss            # the plain submodule form
    assert VQVAEWithBCELoss is losses.VQVAEWithBCELoss
