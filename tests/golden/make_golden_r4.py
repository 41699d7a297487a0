#!/usr/bin/env python3
"""Round-4 fixture, made by running the REFERENCE itself on CPU (authoring container only: needs /root/reference).

    python tests/golden/make_golden_r4.py            # writes vq_img256_dec_bwd.npz next to this file

``vq_img256_dec_bwd.npz`` -- the same run as ``vq_img256.npz`` / ``vq_img256_bwd.npz`` (conf/img_config.yaml model block, 256x256,
B=1, seed 1, loss L1 + q_loss; the script asserts its z / z_q / loss equal the committed fixture's) with what a decoder-backward
parity test needs: ``drec`` = dL/drec at the decoder's output (reference train.py:84-97 with the L1 term of
losses/loss_img.py:79), ``dzq`` = dL/dz_q at ``post_quant_conv``'s input (the data gradient that leaves the decoder,
models/vqvae.py:26-29), and the reference's gradients of a spread of decoder parameters (r4_spec.DEC_GRADS).  Feeding the reference's
z_q into our decoder and the reference's drec into its backward isolates the decoder's kernels -- at B=1 and with the image replicated
(the multi-tile launches bench.py times).  ``refbf16_*``: how far the reference's OWN decoder gradients move under
``torch.autocast("cpu", bfloat16)`` (same z_q, same drec) -- the yardstick for bf16 storage through 29 layers.
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
from oracle.vq_oracle import synth_state_dict, synth_image_batch  # noqa: E402
from r4_spec import DEC_GRADS  # noqa: E402

IMG = dict(ddconfig=dict(z_channels=256, in_channels=3, out_channels=3,
                         channels=[128, 128, 128, 256, 512, 512], num_res_blocks=2, resolution=512,
                         attn_resolutions=[32], dropout=0.0),
           n_embed=8192, embed_dim=256, init_steps=3000, reservoir_size=12500)


def main():
    old = np.load(os.path.join(HERE, "vq_img256.npz"))
    x = synth_image_batch(1, 3, 256, seed=1)
    model = VQBASE(**IMG)
    model.load_state_dict(synth_state_dict(IMG["ddconfig"], IMG["n_embed"], IMG["embed_dim"], seed=1, codebook_scale=1.0), strict=True)
    model.train(True)
    model.quantize.q_counter = model.quantize.q_re_end
    taps = {}

    def keep_zq(m, args):                               # the tensor post_quant_conv receives = z_q (straight-through)
        args[0].retain_grad()
        taps["zq"] = args[0]
    model.post_quant_conv.register_forward_pre_hook(keep_zq)
    model.quant_conv.register_forward_hook(lambda m, i, o: taps.__setitem__("z", o.detach()))
    rec, q_loss = model(x)
    rec.retain_grad()
    loss = (x - rec).abs().mean() + q_loss
    loss.backward()
    assert np.array_equal(taps["z"].numpy(), old["z"]) and np.array_equal(taps["zq"].detach().numpy(), old["z_q"]) \
        and float(loss) == float(old["loss"]), "not the run vq_img256.npz records"
    names = dict(model.named_parameters())
    out = {"drec": rec.grad.numpy().copy(), "dzq": taps["zq"].grad.numpy().copy(), "rec_sub": rec.detach().numpy()[:, :, ::8, ::8].copy()}
    for k, sl in DEC_GRADS.items():
        out["grad:" + k] = names[k].grad.numpy()[sl].copy()
    dec_norm = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for n, p in model.named_parameters()
                           if (n.startswith("decoder.") or n.startswith("post_quant_conv.")) and p.grad is not None))
    # yardstick: the reference's own decoder under torch.autocast(bfloat16) on CPU, same z_q, same drec
    model.zero_grad(set_to_none=True)
    zq = torch.from_numpy(old["z_q"]).requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        rb = model.decode(zq)
    rb.backward(torch.from_numpy(out["drec"]).to(rb.dtype))
    for k, sl in DEC_GRADS.items():
        gb, gf = names[k].grad.double().numpy()[sl], out["grad:" + k].astype(np.float64)
        out["refbf16_l2:" + k] = np.linalg.norm(gb - gf) / np.linalg.norm(gf)
        out["refbf16_max:" + k] = np.abs(gb - gf).max() / np.abs(gf).max()
        print("  reference bf16-autocast vs fp32  %-40s rel-L2 %.3e max-rel %.3e" % (k, out["refbf16_l2:" + k], out["refbf16_max:" + k]))
    gb, gf = zq.grad.double().numpy(), out["dzq"].astype(np.float64)
    out["refbf16_l2:dzq"] = np.linalg.norm(gb - gf) / np.linalg.norm(gf)
    out["refbf16_max:dzq"] = np.abs(gb - gf).max() / np.abs(gf).max()
    print("  reference bf16-autocast vs fp32  %-40s rel-L2 %.3e max-rel %.3e" % ("dL/dz_q", out["refbf16_l2:dzq"], out["refbf16_max:dzq"]))
    np.savez_compressed(os.path.join(HERE, "vq_img256_dec_bwd.npz"), gradnorm_decoder=dec_norm, torch_version=torch.__version__, **out)
    print("vq_img256_dec_bwd.npz: |drec| max %.3e, decoder gradient norm %.5f" % (float(np.abs(out["drec"]).max()), dec_norm))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
