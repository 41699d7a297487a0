"""Which sub-sampled gradients the round-3 fixtures hold (shared by make_golden_r3.py, which writes them from the reference's
run, and tests/test_gpu_parity_r3.py, which cuts the same slices out of our gradients)."""
import numpy as np

# vq_img256_bwd.npz: encoder / quant_conv parameters (name -> slice that keeps the fixture small)
ENC_GRADS = {
    "encoder.model.0.weight": np.s_[:],                      # 3 -> 128 @256^2 (first layer: the whole backward chain above it)
    "encoder.model.0.bias": np.s_[:],
    "encoder.model.1.norm1.weight": np.s_[:],                # GroupNorm affine @256^2
    "encoder.model.2.conv2.weight": np.s_[::4],              # 128 -> 128 @256^2 (the dominant shape's weight gradient)
    "encoder.model.4.conv1.weight": np.s_[::4],              # 128 -> 128 @128^2
    "encoder.model.3.conv.weight": np.s_[::4],               # Downsample 256 -> 128 (stride 2)
    "encoder.model.7.nin_shortcut.weight": np.s_[:],         # 1x1 128 -> 256
    "encoder.model.11.conv1.weight": np.s_[::64],            # 512 -> 512 @32^2
    "encoder.model.14.q.weight": np.s_[::8],                 # AttnBlock @16^2
    "encoder.model.22.weight": np.s_[::32],                  # conv_out 512 -> 256
    "quant_conv.0.weight": np.s_[::4],
}
# transformer_w1024.npz
TR1024 = dict(num_layers=2, hidden_dim=1024, num_attn_heads=16, image_vocab_size=8192, seg_vocab_size=256,
              text_vocab_size=49408 + 256, image_tokens_per_dim=32, seg_tokens_per_dim=16, text_length=256)
TR_GRADS = {
    "transformer.layers.0.attn.qkv.weight": np.s_[::48, ::16],
    "transformer.layers.0.attn.qkv.bias": np.s_[:],
    "transformer.layers.1.mlp.lin1.weight": np.s_[::64, ::16],
    "transformer.layers.1.mlp.lin2.bias": np.s_[:],
    "transformer.layers.0.ln_in.weight": np.s_[:],
    "transformer.layers.0.first_ln_sandwich.weight": np.s_[:],
    "transformer.layers.1.second_ln_sandwich.bias": np.s_[:],
    "transformer.layers.1.attn.out_proj.weight": np.s_[::16, ::16],
    "to_logits.1.weight": np.s_[::128, ::16],
    "image_token_embedding.weight": np.s_[::128, ::16],
}
LOGITS_SUB = np.s_[:, ::8, ::32]
