#!/usr/bin/env python3
"""Round-6 fixtures, produced by running the REFERENCE itself on CPU (authoring container only: needs /root/reference).

  loss_seg.npz      -- the reference's losses/loss_seg.py (BCELossWithQuant, VQVAEWithBCELoss; loaded by file path: the package's
                       __init__ pulls torchvision) on seeded logits / targets: loss values and the gradient w.r.t. the prediction.
  codebook_b32.npz  -- the reference's Codebook.forward (models/modules.py:501-517) at the BENCHED size: 32 x 16 x 16 = 8192 fp32
                       latents against 8192 codes of the post-k-means-like scale (VERDICT r5 "next" #3a).  Inputs are regenerated
                       from the numpy seed by the test; the fixture holds the 8192 indices, the loss, a slice of z_q and the
                       top-2 distance gaps.

    python tests/golden/make_golden_r6.py [loss_seg|codebook_b32]
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def loss_seg_inputs(seed=21, n=2, c=159, hw=8):
    """shared with tests/test_losses_host.py: logits, {0,1} targets (the five heavy channels populated), a q_loss scalar"""
    rs = np.random.RandomState(seed)
    pred = (2.0 * rs.randn(n, c, hw, hw)).astype(np.float32)
    target = (rs.rand(n, c, hw, hw) < 0.3).astype(np.float32)
    qloss = np.float32(0.37)
    return pred, target, qloss


def codebook_b32_inputs(seed=13):
    """shared with tests/test_gpu_kernels.py: z [32,256,16,16] and the [8192,256] codebook, both N(0,1) fp32"""
    rs = np.random.RandomState(seed)
    z = rs.randn(32, 256, 16, 16).astype(np.float32)
    cb = rs.randn(8192, 256).astype(np.float32)
    return z, cb


def make_loss_seg():
    spec = importlib.util.spec_from_file_location("ref_loss_seg", "/root/reference/losses/loss_seg.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    pred, target, qloss = loss_seg_inputs()
    out = {}
    for name in ("BCELossWithQuant", "VQVAEWithBCELoss"):
        for cw in (1.0, 0.25):
            m = getattr(ref, name)(image_channels=159, codebook_weight=cw)
            p = torch.from_numpy(pred).requires_grad_(True)
            loss = m(torch.tensor(qloss), torch.from_numpy(target), p)
            loss.backward()
            out[f"{name}:{cw}:loss"] = loss.detach().numpy()
            out[f"{name}:{cw}:grad"] = p.grad.numpy()
        out[f"{name}:weight"] = m.weight.numpy()
        out[f"{name}:state_keys"] = np.array(sorted(m.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "loss_seg.npz"), torch_version=torch.__version__, **out)


def make_codebook_b32():
    sys.path.insert(0, "/root/reference")
    stub = types.ModuleType("fast_pytorch_kmeans")
    stub.KMeans = object
    sys.modules["fast_pytorch_kmeans"] = stub
    from models.modules import Codebook          # the reference's
    z, cbw = codebook_b32_inputs()
    cb = Codebook(8192, 256, beta=0.25, init_steps=3000, reservoir_size=12500)
    cb.embedding.weight.data = torch.from_numpy(cbw)
    cb.eval()
    zt = torch.from_numpy(z)
    with torch.no_grad():
        zq, loss, idx = cb(zt)
        zf = zt.permute(0, 2, 3, 1).reshape(-1, 256)
        d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(cb.embedding.weight ** 2, dim=1) - 2 * zf @ cb.embedding.weight.t()
        top2 = torch.topk(d, 2, dim=1, largest=False).values
    np.savez_compressed(os.path.join(HERE, "codebook_b32.npz"), idx=idx.numpy().astype(np.int16), loss=loss.numpy(),
                        zq_sub=zq.numpy()[::8, ::16], gap=(top2[:, 1] - top2[:, 0]).numpy().astype(np.float32),
                        torch_version=torch.__version__)


if __name__ == "__main__":
    torch.set_num_threads(8)
    what = sys.argv[1:] or ["loss_seg", "codebook_b32"]
    if "loss_seg" in what:
        make_loss_seg()
    if "codebook_b32" in what:
        make_codebook_b32()
    print("written:", what)
