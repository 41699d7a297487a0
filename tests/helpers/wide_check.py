#!/usr/bin/env python3
"""Runs in a child process of tests/test_gpu_wide.py with MAS_CONV_WIDE_MIN_TILES_PER_CU=0 (read once per process), so that
SMALL shapes take the wide 3x3 kernel (conv3x3_wide.hip: 16x32-pixel tiles, 32-channel chunks, K32 weight image) and can be
compared with a CPU fp32 convolution of the same bf16-rounded operands.  Prints one line per case; exits non-zero on a mismatch.

`wide_check.py multi` (run with MAS_CONV_WGS_PER_CU=1: the grid is then ONE work-group per CU): shapes with MORE tiles than
work-groups, so every work-group really walks 2-3 tiles through the persistent loop bench.py's B=32 launches run (next-tile plan,
LDS-parked output offsets, double-buffered scale/shift table, cross-tile patch / weight DMA, the vmcnt(32) wait for the previous
tile's stores).  The launch geometry is asserted (tiles > grid), not assumed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import mas_hip  # noqa: E402
from mas_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
CASES = [
    # n, cin, h, w, cout, pad4 (t,b,l,r), upsample, act, residual, transpose (dgrad packing)
    (1, 64, 16, 32, 128, (1, 1, 1, 1), False, 0, False, False),    # one tile, one pair
    (2, 128, 40, 72, 128, (1, 1, 1, 1), False, 0, True, False),    # ragged tiles (40 % 16, 72 % 32), several tiles per work-group
    (3, 256, 20, 36, 128, (1, 1, 1, 1), False, 0, False, False),   # four pairs per tile; 36-wide map (ragged second tile column)
    (2, 128, 33, 47, 256, (1, 1, 1, 1), False, 0, True, False),    # two cout tiles (c0 changes between consecutive tiles), odd sizes
    (1, 512, 16, 32, 512, (1, 1, 1, 1), False, 0, False, False),   # eight pairs, four cout tiles
    (2, 128, 12, 20, 128, (1, 1, 1, 1), True, 0, False, False),    # Upsample fold (24 x 40 output)
    (2, 128, 31, 35, 128, (2, 2, 2, 2), False, 0, False, False),   # pad 2 (the zero-stuffed stride-2 data gradient's geometry)
    (2, 128, 40, 64, 128, (1, 1, 1, 1), False, 2, False, False),   # GroupNorm+SiLU prologue
    (3, 256, 20, 36, 128, (1, 1, 1, 1), False, 2, True, False),    # prologue, four pairs, residual
    (2, 128, 18, 34, 256, (1, 1, 1, 1), False, 1, False, False),   # affine-only prologue, two cout tiles
    (40, 128, 32, 32, 128, (1, 1, 1, 1), False, 2, True, False),   # 80 tiles = 80 work-groups, ONE tile each (the multi-tile walk is MULTI_CASES' job)
    (40, 128, 32, 32, 128, (1, 1, 1, 1), False, 0, False, False),
    (2, 256, 32, 64, 128, (1, 1, 1, 1), False, 0, False, True),    # data-gradient packing (in/out swapped, taps flipped): Cout 256 -> Cin 128
]

# more tiles than work-groups (grid = 1 work-group per CU under MAS_CONV_WGS_PER_CU=1): tiles = n * ceil(ho/16) * ceil(wo/32) * cout/128
MULTI_CASES = [
    (40, 128, 64, 96, 128, (1, 1, 1, 1), False, 2, True, False),   # 480 tiles; consecutive tiles of a work-group lie in DIFFERENT images: ss_stage(nt.n, ss_sel ^ 1)
    (40, 128, 64, 96, 128, (1, 1, 1, 1), False, 0, False, False),  # 480 tiles, plain
    (24, 128, 32, 48, 128, (1, 1, 1, 1), True, 0, False, False),   # Upsample fold, 288 tiles
    (8, 128, 64, 96, 384, (1, 1, 1, 1), False, 2, False, False),   # three cout tiles: c0 changes between the tiles a work-group walks (256 % 3 != 0)
    (8, 128, 60, 90, 384, (1, 1, 1, 1), False, 0, True, False),    # the same with ragged tile rows / columns and the residual
    (2, 128, 272, 512, 128, (1, 1, 1, 1), False, 2, True, False),  # 272 tiles per image: consecutive tiles in the SAME image for some work-groups, a new image for others; 3 tiles for 32 of them
    (24, 256, 64, 96, 128, (1, 1, 1, 1), False, 1, False, True),   # data-gradient packing + affine-only prologue, four chunk pairs, 288 tiles
    (24, 128, 62, 94, 128, (2, 2, 2, 2), False, 0, False, False),  # pad 2 (64 x 96 output), 288 tiles
]


def n_tiles(case):
    n, cin, h, w, cout, pad4, ups, act, has_res, tr = case
    hl, wl = (2 * h, 2 * w) if ups else (h, w)
    ho, wo = hl + pad4[0] + pad4[1] - 2, wl + pad4[2] + pad4[3] - 2
    return n * ((ho + 15) // 16) * ((wo + 31) // 32) * (cout // 128)


def silu(u):
    return u * torch.sigmoid(u)


def main():
    bad = 0
    multi = len(sys.argv) > 1 and sys.argv[1] == "multi"
    cases = MULTI_CASES if multi else CASES
    if multi:
        # launch geometry (conv3x3_wide.hip launch_wide): grid = min(tiles, MAS_CONV_WGS_PER_CU * CUs)
        wgs = int(os.environ.get("MAS_CONV_WGS_PER_CU", "0"))
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert wgs == 1, "run `wide_check.py multi` with MAS_CONV_WGS_PER_CU=1"
        grid = wgs * cus
        for case in cases:
            assert n_tiles(case) > grid, (case, n_tiles(case), grid)
        print(f"multi-tile mode: grid {grid} work-groups, tiles per case {[n_tiles(c) for c in cases]}", flush=True)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    for case in cases:
        n, cin, h, w, cout, pad4, ups, act, has_res, tr = case
        g = torch.Generator(device="cpu").manual_seed(hash(case) % 2**31)
        x = torch.randn(n, cin, h, w, generator=g).bfloat16()
        # forward: weight [cout, cin]; transpose: the parameter is [cin_of_fwd = cout here ... ] -- build the EFFECTIVE forward filter
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).bfloat16().float()
        b = 0.1 * torch.randn(cout, generator=g)
        ss = torch.stack([1.0 + 0.2 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)], dim=-1).contiguous()
        hl, wl = (2 * h, 2 * w) if ups else (h, w)
        t, bo, l, r = pad4
        ho, wo = hl + t + bo - 2, wl + l + r - 2
        res = torch.randn(n, cout, ho, wo, generator=g).bfloat16() if has_res else None
        a = x.float()
        if act:
            a = a * ss[..., 0][:, :, None, None] + ss[..., 1][:, :, None, None]
            if act == 2:
                a = silu(a)
            a = a.bfloat16().float()
        if ups:
            a = F.interpolate(a, scale_factor=2.0, mode="nearest")
        ref = F.conv2d(F.pad(a, (l, r, t, bo)), wt, b)
        if has_res:
            ref = ref + res.float()
        cl = lambda z: z.to(dev).contiguous(memory_format=torch.channels_last)
        if tr:   # the parameter whose transpose=1 packing IS `wt`: P[o'][i'][kh][kw] = wt[i'][o'][2-kh][2-kw]
            param = wt.permute(1, 0, 2, 3).flip(2, 3).contiguous()
        else:
            param = wt
        d = ops._desc(n, h, w, cin, ho, wo, cout, 3, 1, t, l, torch.bfloat16, torch.bfloat16, act, ups)
        lay = ops._preferred_layout(d)
        y = ops.conv_fwd_raw(cl(x), ss.to(dev) if act else None, ops.ConvWeight(param.to(dev), tr), b.to(dev), cl(res) if has_res else None,
                             n, h, w, cin, ho, wo, cout, 3, 1, t, l, act, ups, torch.bfloat16)
        torch.cuda.synchronize()
        err = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
        ok = err < 1e-2 and lay == mas_hip.WLAYOUT_K32
        extra = ""
        bad += not ok
        print(("ok   " if ok else "FAIL ") + f"{case}: max-rel {err:.3e} layout {'K32 (wide kernel)' if lay else 'K64 (NOT the wide kernel)'}" + extra, flush=True)
    # determinism: the same launch twice is bitwise identical
    y2 = ops.conv_fwd_raw(cl(x), None, ops.ConvWeight(param.to(dev), tr), b.to(dev), None, n, h, w, cin, ho, wo, cout, 3, 1, t, l, 0, ups, torch.bfloat16)
    y3 = ops.conv_fwd_raw(cl(x), None, ops.ConvWeight(param.to(dev), tr), b.to(dev), None, n, h, w, cin, ho, wo, cout, 3, 1, t, l, 0, ups, torch.bfloat16)
    if not torch.equal(y2, y3):
        print("FAIL run-to-run determinism")
        bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
