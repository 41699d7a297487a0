"""Fused GroupNorm statistics of the wide kernel's epilogue (mas_conv_fwd_stats) against the stand-alone pass (mas_gn_stats), for the
launch geometry of the calling environment -- tests/test_gpu_parity_r4.py runs it under MAS_CONV_WGS_PER_CU=1 (one work-group per CU:
every work-group walks 4-16 tiles, the statistics flush of one tile overlaps the next tile's first stage).  Exit code 1 on a mismatch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
from mas_hip import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(11)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    per_cu = int(os.environ.get("MAS_CONV_WGS_PER_CU", "4") or 4)
    bad = 0
    ops._stats_state["on"] = True
    # n, cin, h, w, cout, residual: whole tiles / ragged rows and columns / two cout tiles / the benched shape
    for (n, c, h, w, cout, res) in ((32, 128, 128, 128, 128, True), (24, 128, 104, 136, 128, False), (16, 128, 96, 160, 256, True),
                                    (32, 128, 256, 256, 128, True)):
        x = torch.randn(n, c, h, w, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(cout, c, 3, 3, device=dev, generator=g) / (9 * c) ** 0.5
        b = 0.1 * torch.randn(cout, device=dev, generator=g)
        r = torch.randn(n, cout, h, w, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last) if res else None
        y, part, rows = ops.conv_fwd_raw(x, None, ops.ConvWeight(wt, False), b, r, n, h, w, c, h, w, cout, 3, 1, 1, 1, 0, False, bf, want_stats=True)
        kern = ops.last_kernel()
        tiles = n * ((h + 15) // 16) * ((w + 31) // 32) * (cout // 128)
        grid = min(tiles, per_cu * cus)
        ga, be = 1.0 + 0.1 * torch.randn(cout, device=dev, generator=g), 0.1 * torch.randn(cout, device=dev, generator=g)
        ok = part is not None and kern == "conv3x3_wide" and tiles >= 3 * grid
        if ok:
            mr_f, ss_f = ops.gn_stats(y, ga, be, 32, 1e-6, part, rows)
            mr_s, ss_s = ops.gn_stats(y, ga, be, 32, 1e-6)
            e1 = float((mr_f - mr_s).abs().max() / mr_s.abs().max())
            e2 = float((ss_f - ss_s).abs().max() / ss_s.abs().max())
            y2, part2, _ = ops.conv_fwd_raw(x, None, ops.ConvWeight(wt, False), b, r, n, h, w, c, h, w, cout, 3, 1, 1, 1, 0, False, bf, want_stats=True)
            ok = e1 < 1e-4 and e2 < 1e-4 and torch.equal(part, part2) and torch.equal(y, y2)
            msg = f"mean/rstd {e1:.2e} scale/shift {e2:.2e}, bitwise repeatable {torch.equal(part, part2)}"
        else:
            msg = f"kernel {kern}, table {'present' if part is not None else 'ABSENT'}, tiles {tiles} on {grid} work-groups"
        bad += not ok
        print(("ok   " if ok else "FAIL ") + f"n={n} {c}->{cout} {h}x{w} res={res}: {tiles} tiles on {grid} work-groups ({tiles / grid:.1f} per work-group): {msg}", flush=True)
        del x, y, r
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
