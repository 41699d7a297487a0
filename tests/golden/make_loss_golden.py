#!/usr/bin/env python3
"""Generates tests/golden/disc_tiny.npz by running the REFERENCE's own PatchGAN discriminator and loss functions on CPU
(authoring container only).  ``losses/__init__.py`` cannot be imported here (it pulls torchvision through lpips.py:4), so
``losses/discriminator.py`` is loaded by file path -- it only needs torch -- and the three pure functions of loss_img.py
(adopt_weight / hinge_d_loss / vanilla_d_loss, :11-31) are exec'd from its source text, nothing restated.
Weights: oracle.loss_oracle.synth_disc_state_dict, loaded with ``load_state_dict(strict=True)`` (= key / shape proof)."""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.loss_oracle import synth_disc_state_dict  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_discriminator", "/root/reference/losses/discriminator.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

src = open("/root/reference/losses/loss_img.py").read()
ns = {"torch": torch, "F": torch.nn.functional}
for node in ast.parse(src).body:                      # only the three pure functions, verbatim
    if isinstance(node, ast.FunctionDef) and node.name in ("adopt_weight", "hinge_d_loss", "vanilla_d_loss"):
        exec(compile(ast.Module([node], []), "loss_img.py", "exec"), ns)


def main():
    sd = synth_disc_state_dict(seed=7)
    d = ref.Discriminator()
    d.load_state_dict(sd, strict=True)
    rs = np.random.RandomState(11)
    real = torch.from_numpy(rs.rand(2, 3, 64, 64).astype(np.float32))
    fake = torch.from_numpy(np.clip(real.numpy() + 0.15 * rs.randn(2, 3, 64, 64), 0, 1).astype(np.float32)).requires_grad_(True)
    out = {}
    d.eval()                                       # eval-mode logits with the synthetic running statistics, before training touches them
    with torch.no_grad():
        out["logits_real_eval"] = d(real).numpy()
    d.train()
    lr, lf = d(real), d(fake)
    out["logits_real"], out["logits_fake"] = lr.detach().numpy(), lf.detach().numpy()
    d_loss = ns["hinge_d_loss"](lr, lf)
    out["hinge"] = d_loss.detach().numpy()
    out["vanilla"] = ns["vanilla_d_loss"](lr, lf).detach().numpy()
    g_loss = -torch.mean(lf)
    out["g_loss"] = g_loss.detach().numpy()
    (gin,) = torch.autograd.grad(g_loss, fake, retain_graph=True)
    out["grad_fake:g_loss"] = gin.numpy()
    d.zero_grad()
    d_loss.backward()
    for k in ("model.0.weight", "model.0.bias", "model.2.weight", "model.3.weight", "model.3.bias", "model.5.weight", "model.8.weight",
              "model.9.bias", "model.11.weight", "model.11.bias"):
        gk = dict(d.named_parameters())[k].grad.numpy()
        out["grad:" + k] = gk[::8, ::8] if gk.size > 200000 else gk        # big conv gradients: every 8th filter / input channel
    out["running_mean:model.3"] = d.model[3].running_mean.numpy().copy()      # after the two training-mode forwards above
    out["running_var:model.3"] = d.model[3].running_var.numpy().copy()
    out["adopt"] = np.array([ns["adopt_weight"](0.8, 10, threshold=20), ns["adopt_weight"](0.8, 30, threshold=20)], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "disc_tiny.npz"), torch_version=torch.__version__, real=real.numpy(), fake=fake.detach().numpy(), **out)
    print("wrote disc_tiny.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
