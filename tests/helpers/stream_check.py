#!/usr/bin/env python3
"""Runs in a child process of tests/test_gpu_stream.py with MAS_CONV_STREAM_MIN_TILES_PER_CU=0 (the knob is read once per
process), so that SMALL shapes take the stream-scheduled 3x3 kernel (conv3x3_stream.hip) and can be compared with a CPU
fp32 convolution of the same bf16-rounded operands.  Prints one line per case and exits non-zero on a mismatch.

`stream_check.py multi` (run with MAS_CONV_WGS_PER_CU=1: one work-group per CU): more 16x16-pixel tiles than work-groups, so every
work-group walks several tiles (deferred stores of the previous tile, next-tile plan and prefetch); tiles > grid is asserted."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mas_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
CASES = [
    # n, cin, h, w, cout, pad4 (t,b,l,r), upsample, act, residual
    (1, 128, 16, 16, 128, (1, 1, 1, 1), False, 0, False),     # one tile, one pair
    (2, 128, 40, 24, 128, (1, 1, 1, 1), False, 0, True),      # ragged tiles (40 % 16, 24 % 16), several tiles per work-group
    (3, 256, 20, 36, 128, (1, 1, 1, 1), False, 0, False),     # two pairs per tile
    (2, 128, 33, 17, 256, (1, 1, 1, 1), False, 0, True),      # two cout tiles (c0 changes between consecutive tiles), odd sizes
    (1, 512, 16, 32, 512, (1, 1, 1, 1), False, 0, False),     # four pairs, four cout tiles
    (2, 128, 12, 20, 128, (1, 1, 1, 1), True, 0, False),      # Upsample fold (24 x 40 output)
    (2, 128, 31, 19, 128, (2, 2, 2, 2), False, 0, False),     # pad 2 (the zero-stuffed stride-2 data gradient's geometry)
    (2, 128, 40, 24, 128, (1, 1, 1, 1), False, 2, False),     # GroupNorm+SiLU prologue
    (3, 256, 20, 36, 128, (1, 1, 1, 1), False, 2, True),      # prologue, two pairs, residual
    (2, 128, 18, 18, 256, (1, 1, 1, 1), False, 1, False),     # affine-only prologue, two cout tiles
    (40, 128, 32, 32, 128, (1, 1, 1, 1), False, 2, True),     # 160 tiles = 160 work-groups, ONE tile each (the multi-tile walk is MULTI_CASES' job)
    (40, 128, 32, 32, 128, (1, 1, 1, 1), False, 0, False),
]
# more tiles than work-groups under MAS_CONV_WGS_PER_CU=1: tiles = n * ceil(ho/16) * ceil(wo/16) * cout/128
MULTI_CASES = [
    (40, 128, 48, 64, 128, (1, 1, 1, 1), False, 2, True),     # 480 tiles, prologue + residual, a new image at every step of the walk
    (40, 128, 48, 64, 128, (1, 1, 1, 1), False, 0, False),    # 480 tiles, plain
    (24, 128, 24, 32, 128, (1, 1, 1, 1), True, 0, False),     # Upsample fold (48 x 64 output), 288 tiles
    (8, 128, 48, 64, 384, (1, 1, 1, 1), False, 2, False),     # three cout tiles: c0 changes between the tiles a work-group walks
    (8, 256, 45, 60, 384, (1, 1, 1, 1), False, 0, True),      # ragged tiles, two chunk pairs, residual, 288 tiles
    (1, 128, 272, 272, 128, (1, 1, 1, 1), False, 2, True),    # 289 tiles of ONE image: the walk stays inside an image
]


def n_tiles(case):
    n, cin, h, w, cout, pad4, ups, act, has_res = case
    hl, wl = (2 * h, 2 * w) if ups else (h, w)
    ho, wo = hl + pad4[0] + pad4[1] - 2, wl + pad4[2] + pad4[3] - 2
    return n * ((ho + 15) // 16) * ((wo + 15) // 16) * (cout // 128)


def silu(u):
    return u * torch.sigmoid(u)


def main():
    bad = 0
    multi = len(sys.argv) > 1 and sys.argv[1] == "multi"
    cases = MULTI_CASES if multi else CASES
    if multi:
        wgs = int(os.environ.get("MAS_CONV_WGS_PER_CU", "0"))
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert wgs == 1, "run `stream_check.py multi` with MAS_CONV_WGS_PER_CU=1"
        for case in cases:
            assert n_tiles(case) > wgs * cus, (case, n_tiles(case), wgs * cus)
        print(f"multi-tile mode: grid {wgs * cus} work-groups, tiles per case {[n_tiles(c) for c in cases]}", flush=True)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    for case in cases:
        n, cin, h, w, cout, pad4, ups, act, has_res = case
        g = torch.Generator(device="cpu").manual_seed(hash(case) % 2**31)
        x = torch.randn(n, cin, h, w, generator=g).bfloat16()
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
        wp = ops.pack_conv_weight(wt.to(dev), False, torch.bfloat16)
        y = ops.conv_fwd_raw(cl(x), ss.to(dev) if act else None, wp, b.to(dev), cl(res) if has_res else None,
                             n, h, w, cin, ho, wo, cout, 3, 1, t, l, act, ups, torch.bfloat16)
        torch.cuda.synchronize()
        err = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
        ok = err < 1e-2
        bad += not ok
        print(("ok   " if ok else "FAIL ") + f"{case}: max-rel {err:.3e}", flush=True)
    # determinism: the same launch twice is bitwise identical (no order-dependent accumulation in this kernel)
    y2 = ops.conv_fwd_raw(cl(x), None, wp, b.to(dev), None, n, h, w, cin, ho, wo, cout, 3, 1, t, l, 0, ups, torch.bfloat16)
    y3 = ops.conv_fwd_raw(cl(x), None, wp, b.to(dev), None, n, h, w, cin, ho, wo, cout, 3, 1, t, l, 0, ups, torch.bfloat16)
    if not torch.equal(y2, y3):
        print("FAIL run-to-run determinism")
        bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
