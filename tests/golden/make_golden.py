#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE itself on CPU.

Run only in the authoring container (needs /root/reference; it does not exist on the
GPU box).  The reference is imported unmodified; the single absent third-party import
(`fast_pytorch_kmeans`, used only by the k-means re-init branch models/modules.py:489-499,
which is not entered) is stubbed.  Weights come from oracle.vq_oracle.synth_state_dict
(numpy RandomState => independent of torch's RNG stream) and are loaded with
``load_state_dict(strict=True)``, which also proves state_dict key/shape equality.

    python tests/golden/make_golden.py            # writes the fixtures next to this file
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
_stub = types.ModuleType("fast_pytorch_kmeans")
_stub.KMeans = object
sys.modules["fast_pytorch_kmeans"] = _stub

from models import VQBASE  # noqa: E402  (the reference's)
from models.transformer import MakeAScene  # noqa: E402
from oracle.vq_oracle import synth_state_dict, synth_image_batch  # noqa: E402
from oracle.transformer_oracle import synth_transformer_state_dict, synth_tokens  # noqa: E402

TINY = dict(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64, 64],
                          num_res_blocks=1, resolution=32, attn_resolutions=[8], dropout=0.0),
            n_embed=64, embed_dim=32, init_steps=3000, reservoir_size=12500)
IMG = dict(ddconfig=dict(z_channels=256, in_channels=3, out_channels=3,
                         channels=[128, 128, 128, 256, 512, 512], num_res_blocks=2, resolution=512,
                         attn_resolutions=[32], dropout=0.0),
           n_embed=8192, embed_dim=256, init_steps=3000, reservoir_size=12500)
# BASELINE config 1: the model block of conf/seg_config.yaml VERBATIM (ch / ch_mult / out_ch / double_z are swallowed by
# **kwargs, modules.py:217,338, so the decoder emits 3 channels), n_embed 256 as BASELINE states, plus the two kwargs the
# YAML lacks.  SEG_EFF = the constructor arguments that take effect (for the synthetic weights).
SEG_YAML = dict(embed_dim=256, n_embed=256, init_steps=3000, reservoir_size=12500,
                ddconfig=dict(double_z=False, z_channels=256, resolution=256, in_channels=159, out_ch=159, ch=128,
                              ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0))
SEG_EFF = dict(z_channels=256, in_channels=159, out_channels=3, channels=[128, 128, 128, 256, 512, 512], num_res_blocks=2,
               resolution=256, attn_resolutions=[16], dropout=0.0)
GRAD_KEYS_TINY = ["encoder.model.0.weight", "encoder.model.1.norm1.weight", "encoder.model.1.conv2.bias",
                  "encoder.model.3.nin_shortcut.weight", "encoder.model.6.q.weight", "encoder.model.8.norm.bias",
                  "quant_conv.0.weight", "quant_conv.1.weight", "quantize.embedding.weight",
                  "post_quant_conv.weight", "decoder.model.2.proj_out.weight", "decoder.model.8.conv.weight",
                  "decoder.model.16.weight"]


def run_vq(cfg, x, seed, scale, train=True, grads=(), eff=None, loss_fn=None):
    model = VQBASE(**cfg)
    sd = synth_state_dict(eff or cfg["ddconfig"], cfg["n_embed"], cfg["embed_dim"], seed=seed, codebook_scale=scale)
    model.load_state_dict(sd, strict=True)
    model.train(train)
    model.quantize.q_counter = model.quantize.q_re_end  # steady state: VQ active, no k-means
    taps = {}
    model.encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("h", o.detach()))
    model.quant_conv.register_forward_hook(lambda m, i, o: taps.__setitem__("z", o.detach()))
    model.quantize.register_forward_hook(lambda m, i, o: taps.__setitem__("q", o))
    out = {}
    if train:
        rec, q_loss = model(x)
        loss = (loss_fn(x, rec) if loss_fn else (x - rec).abs().mean()) + q_loss
        loss.backward()
        out["loss"] = loss.detach().numpy()
        names = dict(model.named_parameters())
        for k in grads:
            out["grad:" + k] = names[k].grad.numpy().copy()
        out["gradnorm_total"] = np.sqrt(sum(float((p.grad.double() ** 2).sum())
                                            for p in model.parameters() if p.grad is not None))
    else:
        with torch.no_grad():
            rec, q_loss = model(x)
    out.update(rec=rec.detach().numpy(), q_loss=q_loss.detach().numpy(), h=taps["h"].numpy(),
               z=taps["z"].numpy(), z_q=taps["q"][0].detach().numpy(),
               idx=taps["q"][2].numpy().astype(np.int64))
    return out


def seg128():
    """BASELINE configs[0]: VQ-SEG 128x128, codebook 256, batch 4, conf/seg_config.yaml, on the reference's CPU path."""
    x = synth_image_batch(4, 159, 128, seed=4)
    sg = run_vq(SEG_YAML, x, seed=4, scale=1.0, train=True, eff=SEG_EFF, loss_fn=lambda x, rec: rec.abs().mean(),
                grads=["encoder.model.0.weight", "decoder.model.28.weight", "quantize.embedding.weight"])
    np.savez_compressed(os.path.join(HERE, "vq_seg128.npz"), rec_sub=sg["rec"][:, :, ::4, ::4], q_loss=sg["q_loss"], loss=sg["loss"],
                        idx=sg["idx"], z_sub=sg["z"][:, ::4], gradnorm_total=sg["gradnorm_total"],
                        **{k: (v[:, ::8] if v.ndim == 4 and v.shape[1] > 64 else v) for k, v in sg.items() if k.startswith("grad:")})


def img256():
    """full VQ-IMG 256^2 (conf/img_config.yaml model block), B=1, fwd+bwd: sub-sampled activations (fixture size) plus the
    FULL latents z / z_q (the bf16 parity test feeds the reference's z_q to our decoder and compares latents end to end)."""
    xi = synth_image_batch(1, 3, 256, seed=1)
    im = run_vq(IMG, xi, seed=1, scale=1.0, train=True, grads=["decoder.model.28.weight", "encoder.model.0.weight"])
    np.savez_compressed(os.path.join(HERE, "vq_img256.npz"), rec_sub=im["rec"][:, :, ::8, ::8], q_loss=im["q_loss"],
                        loss=im["loss"], idx=im["idx"], z_sub=im["z"][:, ::8], h_sub=im["h"][:, ::8], z=im["z"], z_q=im["z_q"],
                        gradnorm_total=im["gradnorm_total"], **{k: v for k, v in im.items() if k.startswith("grad:")})


def main():
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "seg128":          # only the config-1 fixture
        seg128()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "img256":          # only the config-2 (B=1) fixture
        img256()
        return
    seg128()
    # ---- tiny VQ: full tensors, train + eval, fwd + bwd ------------------------
    x = synth_image_batch(2, 3, 32, seed=0)
    tr = run_vq(TINY, x, seed=0, scale=1.0, train=True, grads=GRAD_KEYS_TINY)
    ev = run_vq(TINY, x, seed=0, scale=1.0, train=False)
    np.savez_compressed(os.path.join(HERE, "vq_tiny.npz"), torch_version=torch.__version__,
                        **{"train:" + k: v for k, v in tr.items()}, **{"eval:" + k: v for k, v in ev.items()})
    # ---- VQ-SEG-like tiny (159 input channels, out_channels=159) ----------------
    seg = dict(TINY, ddconfig=dict(TINY["ddconfig"], in_channels=159, out_channels=159))
    xs = synth_image_batch(2, 159, 16, seed=3)
    sg = run_vq(seg, xs, seed=3, scale=1.0, train=True, grads=["encoder.model.0.weight", "decoder.model.16.weight"])
    np.savez_compressed(os.path.join(HERE, "vq_seg_tiny.npz"), **{"train:" + k: v for k, v in sg.items()})
    img256()
    # ---- codebook lookup alone (the bit-exact gate), default-init + scaled -------
    rs = np.random.RandomState(7)
    z = torch.from_numpy(rs.randn(4, 256, 16, 16).astype(np.float32))
    from models.modules import Codebook
    for tag, cbw in (("scaled", rs.randn(8192, 256).astype(np.float32)),
                     ("default", rs.uniform(-1 / 8192, 1 / 8192, size=(8192, 256)).astype(np.float32))):
        cb = Codebook(8192, 256, beta=0.25, init_steps=3000, reservoir_size=12500)
        cb.embedding.weight.data = torch.from_numpy(cbw)
        cb.eval()
        zq, loss, idx = cb(z)
        d = (torch.sum(z.permute(0, 2, 3, 1).reshape(-1, 256) ** 2, dim=1, keepdim=True)
             + torch.sum(cb.embedding.weight ** 2, dim=1) - 2 * z.permute(0, 2, 3, 1).reshape(-1, 256) @ cb.embedding.weight.t())
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        np.savez_compressed(os.path.join(HERE, f"codebook_{tag}.npz"), idx=idx.numpy(), loss=loss.detach().numpy(),
                            zq_sub=zq.detach().numpy()[:, ::16], gap=(top2[:, 1] - top2[:, 0]).detach().numpy())
    # ---- tiny transformer ---------------------------------------------------------
    tcfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40,
                text_vocab_size=50 + 8, image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    m = MakeAScene(**tcfg)
    m.device = torch.device("cpu")  # reference never sets it (transformer.py:332,352)
    tsd = synth_transformer_state_dict(tcfg, seed=5)
    m.load_state_dict(tsd, strict=True)
    text, segt, imgt = synth_tokens(tcfg, batch=2, seed=5)
    logits = m(text, segt, imgt)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), imgt.reshape(-1))
    loss.backward()
    names = dict(m.named_parameters())
    np.savez_compressed(os.path.join(HERE, "transformer_tiny.npz"), logits=logits.detach().numpy(),
                        loss=loss.detach().numpy(),
                        **{"grad:" + k: names[k].grad.numpy() for k in
                           ["transformer.layers.0.attn.qkv.weight", "transformer.layers.1.mlp.lin2.bias",
                            "text_token_embedding.weight", "to_logits.1.weight"]})
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
