"""``VQBASE`` behind the reference's surface (reference models/vqvae.py:8-39): same constructor
(``ddconfig, n_embed, embed_dim, init_steps, reservoir_size``), same submodule names and
``state_dict`` keys (348 entries for conf/img_config.yaml), same ``encode/decode/decode_code/forward``.
encoder -> quant_conv (1x1 conv + SyncBatchNorm) -> Codebook -> post_quant_conv -> decoder, every
convolution / norm (GroupNorm and, since round 5, the SyncBatchNorm) / lookup on the hand-written gfx950 kernels of libmas_hip.so."""
import torch
from torch import nn

from .modules import Codebook, Conv2d, Decoder, Encoder, SyncBatchNorm


class VQBASE(nn.Module):
    def __init__(self, ddconfig, n_embed, embed_dim, init_steps, reservoir_size):
        super(VQBASE, self).__init__()
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quantize = Codebook(n_embed, embed_dim, beta=0.25, init_steps=init_steps, reservoir_size=reservoir_size)
        # the latent tail is stored fp32 end to end: indices come from fp32 z
        qc = Conv2d(ddconfig["z_channels"], embed_dim, 1)
        qc.in_dtype = qc.out_dtype = torch.float32
        self.quant_conv = nn.Sequential(qc, SyncBatchNorm(embed_dim))
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.post_quant_conv.in_dtype = self.post_quant_conv.out_dtype = torch.float32

    def encode(self, x):
        h = self.encoder(x)
        h = self.quant_conv(h)
        quant, emb_loss, info = self.quantize(h)
        return quant, emb_loss

    @torch.no_grad()
    def encode_to_indices(self, x):
        """Frozen-VQ tokenisation for stage 2: images -> codebook indices [B, h*w] (int64), row-major over
        the latent grid -- what ``train.py:141-145`` consumes as ``img_token`` / ``seg_token``.  The reference's ``encode``
        drops the indices (vqvae.py:23-24); this is the same encoder -> quant_conv -> Codebook lookup, keeping them.  Call in
        ``eval()`` mode (SyncBatchNorm running statistics; the codebook warm-up schedule untouched)."""
        if self.training:
            raise RuntimeError("encode_to_indices: tokenise with the frozen model in eval() mode")
        h = self.quant_conv(self.encoder(x))
        _, _, idx = self.quantize(h)
        return idx.view(x.shape[0], -1)

    def decode(self, quant):
        quant = self.post_quant_conv(quant)
        dec = self.decoder(quant)
        return dec

    def decode_code(self, code_b):
        # the reference calls a method that does not exist (vqvae.py:32 `embed_code`); the intended
        # lookup is get_codebook_entry (modules.py:519) on a square token grid
        b, n = code_b.shape[0], code_b.reshape(code_b.shape[0], -1).shape[1]
        side = int(round(n ** 0.5))
        quant_b = self.quantize.get_codebook_entry(code_b.reshape(b, -1), (b, side, side, self.quantize.codebook_dim))
        return self.decode(quant_b)

    def forward(self, input):
        quant, diff = self.encode(input)
        dec = self.decode(quant)
        return dec, diff
