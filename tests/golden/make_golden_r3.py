#!/usr/bin/env python3
"""Round-3 fixtures, made by running the REFERENCE itself on CPU (authoring container only: needs /root/reference).

    python tests/golden/make_golden_r3.py            # writes vq_img256_bwd.npz and transformer_w1024.npz next to this file

* ``vq_img256_bwd.npz`` -- the same run as ``vq_img256.npz`` (conf/img_config.yaml model block, 256x256, B=1, seed 1, loss
  L1 + q_loss; the script asserts its z / loss equal the committed fixture's) with what the encoder-backward parity test needs:
  ``dz`` = dL/dz at the output of ``quant_conv`` (reference models/vqvae.py:21-22), and the reference's gradients of a few
  encoder parameters.  Injecting the REFERENCE's dz into our encoder backward separates kernel error from the codebook-index
  flips bf16 latents cause downstream.  ``refbf16_l2:<key>`` / ``refbf16_max:<key>``: the deviation of the
  reference's OWN encoder gradients under ``torch.autocast("cpu", bfloat16)`` from its fp32 ones, same dz -- the yardstick for
  what bf16 storage costs this 23-layer backward (GroupNorm's mean-subtraction cancels most of each gradient, so rounding
  noise is amplified towards the first layers: 3 % at quant_conv, 10 % at encoder.model.0).
* ``transformer_w1024.npz`` -- MakeAScene (reference models/transformer.py:275-378) at BASELINE config 4's WIDTH: 2 layers,
  d=1024, 16 heads (head_dim 64), 256 text + 256 seg + 1024 image tokens (S=1536), vocabularies of config 4, B=1, fp32:
  sub-sampled logits, the loss and sub-sampled gradients (LN / GELU / Linear / colsum at d=1024, rows=1536).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
_stub = types.ModuleType("fast_pytorch_kmeans")
_stub.KMeans = object
sys.modules["fast_pytorch_kmeans"] = _stub

from models import VQBASE  # noqa: E402  (the reference's)
from models.transformer import MakeAScene  # noqa: E402
from oracle.vq_oracle import synth_state_dict, synth_image_batch  # noqa: E402
from oracle.transformer_oracle import synth_transformer_state_dict, synth_tokens  # noqa: E402

IMG = dict(ddconfig=dict(z_channels=256, in_channels=3, out_channels=3,
                         channels=[128, 128, 128, 256, 512, 512], num_res_blocks=2, resolution=512,
                         attn_resolutions=[32], dropout=0.0),
           n_embed=8192, embed_dim=256, init_steps=3000, reservoir_size=12500)
from r3_spec import ENC_GRADS, LOGITS_SUB, TR1024, TR_GRADS  # noqa: E402


def img256_bwd():
    old = np.load(os.path.join(HERE, "vq_img256.npz"))
    x = synth_image_batch(1, 3, 256, seed=1)
    model = VQBASE(**IMG)
    model.load_state_dict(synth_state_dict(IMG["ddconfig"], IMG["n_embed"], IMG["embed_dim"], seed=1, codebook_scale=1.0), strict=True)
    model.train(True)
    model.quantize.q_counter = model.quantize.q_re_end
    taps = {}

    def keep_z(m, i, o):
        o.retain_grad()
        taps["z"] = o
    model.quant_conv.register_forward_hook(keep_z)
    rec, q_loss = model(x)
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    z = taps["z"]
    assert np.array_equal(z.detach().numpy(), old["z"]) and float(loss) == float(old["loss"]), "not the run vq_img256.npz records"
    names = dict(model.named_parameters())
    out = {"dz": z.grad.numpy().copy()}
    for k, sl in ENC_GRADS.items():
        out["grad:" + k] = names[k].grad.numpy()[sl].copy()
    enc_norm = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for n, p in model.named_parameters()
                           if (n.startswith("encoder.") or n.startswith("quant_conv.")) and p.grad is not None))
    # yardstick: the REFERENCE's own encoder under torch.autocast(bfloat16) on CPU (bf16 convolutions, fp32 GroupNorm: PyTorch's
    # autocast policy), same weights, same input, same injected dz -- how far a bf16 run of the reference is from its own fp32 run
    model.zero_grad(set_to_none=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        zb = model.quant_conv(model.encoder(x))
    zb.backward(z.grad.to(zb.dtype))
    for k, sl in ENC_GRADS.items():
        gb, gf = names[k].grad.double().numpy()[sl], out["grad:" + k].astype(np.float64)
        out["refbf16_l2:" + k] = np.linalg.norm(gb - gf) / np.linalg.norm(gf)
        out["refbf16_max:" + k] = np.abs(gb - gf).max() / np.abs(gf).max()
        print("  reference bf16-autocast vs fp32  %-36s rel-L2 %.3e max-rel %.3e" % (k, out["refbf16_l2:" + k], out["refbf16_max:" + k]))
    np.savez_compressed(os.path.join(HERE, "vq_img256_bwd.npz"), gradnorm_encoder=enc_norm, torch_version=torch.__version__, **out)
    print("vq_img256_bwd.npz: |dz| max %.3e, encoder gradient norm %.5f" % (float(np.abs(out["dz"]).max()), enc_norm))


def transformer_w1024():
    m = MakeAScene(**TR1024)
    m.device = torch.device("cpu")                     # the reference never sets it (transformer.py:332,352)
    m.load_state_dict(synth_transformer_state_dict(TR1024, seed=9), strict=True)
    text, seg, img = synth_tokens(TR1024, batch=1, seed=9)
    logits = m(text, seg, img)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    names = dict(m.named_parameters())
    np.savez_compressed(os.path.join(HERE, "transformer_w1024.npz"), logits_sub=logits.detach().numpy()[LOGITS_SUB],
                        logits_absmax=float(logits.detach().abs().max()), loss=loss.detach().numpy(),
                        torch_version=torch.__version__,
                        **{"grad:" + k: names[k].grad.numpy()[sl].copy() for k, sl in TR_GRADS.items()})
    print("transformer_w1024.npz: logits", tuple(logits.shape), "loss %.5f" % float(loss))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "img256"):
        img256_bwd()
    if which in ("all", "tr1024"):
        transformer_w1024()
