"""Shapes of tests/test_gpu_stream_th8.py (shared with tests/helpers/stream_th8_out.py, which runs in a child process)."""
import torch

# name, n, cin, cout, h, w, residual, x2 upsample folded into the convolution
CASES = [
    ("level16_512_res", 32, 512, 512, 16, 16, True, False),      # the 16x16 level of VQ-IMG at the benched batch: 128 -> 256 tiles
    ("level16_256_512", 4, 256, 512, 16, 16, False, False),
    ("ragged_13x20", 5, 128, 256, 13, 20, False, False),         # bottom tile 5 rows, right tile 4 columns
    ("seg_8x8_res", 3, 128, 128, 8, 8, True, False),             # VQ-SEG's latent grid: one 8-row tile per image, half the columns idle
    ("mid_24x40", 2, 128, 128, 24, 40, False, False),            # three 8-row tiles x three column tiles
    ("upsample_8_to_16", 2, 128, 128, 8, 8, False, True),
]


def make_case(case):
    name, n, cin, cout, h, w, res, ups = case
    g = torch.Generator().manual_seed(len(name) * 7 + cin + h)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).bfloat16().float()
    b = 0.1 * torch.randn(cout, generator=g)
    ho, wo = (2 * h, 2 * w) if ups else (h, w)
    r = torch.randn(n, cout, ho, wo, generator=g).bfloat16() if res else None
    return x, wt, b, r
