"""Which sub-sampled decoder gradients vq_img256_dec_bwd.npz holds (shared by make_golden_r4.py, which writes them from the
reference's run, and tests/test_gpu_parity_r4.py, which cuts the same slices out of our gradients)."""
import numpy as np

DEC_GRADS = {
    "post_quant_conv.weight": np.s_[::4],                    # 1x1 256 -> 256 @16^2 (fp32 latent tail)
    "decoder.model.0.weight": np.s_[::16],                   # conv_in 256 -> 512 @16^2
    "decoder.model.1.norm1.weight": np.s_[:],                # GroupNorm affine @16^2
    "decoder.model.2.q.weight": np.s_[::8],                  # AttnBlock @16^2
    "decoder.model.10.conv.weight": np.s_[::64],             # Upsample 16 -> 32, 512 -> 512 (folded nearest x2)
    "decoder.model.12.conv2.weight": np.s_[::64],            # 512 -> 512 @32^2
    "decoder.model.15.nin_shortcut.weight": np.s_[::2],      # 1x1 512 -> 256 @64^2
    "decoder.model.15.conv1.weight": np.s_[::32],            # 512 -> 256 @64^2
    "decoder.model.18.conv.weight": np.s_[::16],             # Upsample 64 -> 128, 256 -> 256
    "decoder.model.19.conv1.weight": np.s_[::8],             # 256 -> 128 @128^2
    "decoder.model.22.conv.weight": np.s_[::4],              # Upsample 128 -> 256, 128 -> 128 (the dominant map, folded x2)
    "decoder.model.25.conv2.weight": np.s_[::4],             # 128 -> 128 @256^2 (the dominant shape's weight gradient)
    "decoder.model.25.norm2.weight": np.s_[:],               # GroupNorm affine @256^2
    "decoder.model.26.bias": np.s_[:],                       # norm_out
    "decoder.model.28.weight": np.s_[:],                     # conv_out 128 -> 3 @256^2 (the layer train.py:96 differentiates against)
    "decoder.model.28.bias": np.s_[:],
}
