#!/usr/bin/env python3
"""Generates tests/golden/lpips_tiny.npz by running the REFERENCE's own LPIPS class (losses/lpips.py, loaded by file path) on CPU
(authoring container only).  Two things the reference needs are not available offline and are stubbed, nothing else:
``torchvision.models.vgg16`` (a module stub that builds the published VGG16 'D' ``features`` stack, random-initialised) and the
checkpoint download of ``load_from_pretrained`` (lpips.py:63-65, skipped).  Weights: oracle.lpips_oracle.synth_lpips_state_dict,
loaded with ``load_state_dict(strict=True)`` (= proof of the key layout and shapes)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.lpips_oracle import VGG_CFG, expected_keys, synth_lpips_state_dict  # noqa: E402


def _vgg16(pretrained=False):
    layers, cin = [], 3
    for v in VGG_CFG + ["M"]:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m


tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models")
tvm.vgg16 = _vgg16; tv.models = tvm
sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
spec = importlib.util.spec_from_file_location("ref_lpips", "/root/reference/losses/lpips.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
ref.LPIPS.load_from_pretrained = lambda self, name="vgg_lpips": None


def main():
    torch.manual_seed(0)
    m = ref.LPIPS().eval()
    keys = list(m.state_dict().keys())
    assert keys == expected_keys(), (keys, expected_keys())
    sd = synth_lpips_state_dict(seed=3)
    m.load_state_dict(sd, strict=True)
    rs = np.random.RandomState(5)
    real = torch.from_numpy(rs.rand(2, 3, 32, 32).astype(np.float32))
    fake = torch.from_numpy(np.clip(real.numpy() + 0.2 * rs.randn(2, 3, 32, 32), 0, 1).astype(np.float32)).requires_grad_(True)
    out = m(real, fake)
    out.sum().backward()
    feats = m.vgg(m.scaling_layer(real))
    np.savez_compressed(os.path.join(HERE, "lpips_tiny.npz"), real=real.numpy(), fake=fake.detach().numpy(), out=out.detach().numpy(),
                        dfake=fake.grad.numpy(), keys=np.array(keys),
                        feat_means=np.array([float(f.mean()) for f in feats], dtype=np.float64),
                        feat_shapes=np.array([list(f.shape) for f in feats]))
    print("lpips", out.flatten().tolist(), "grad norm", float(fake.grad.norm()), "torch", torch.__version__)


if __name__ == "__main__":
    main()
