#!/usr/bin/env python3
"""`Upsample` + conv in its sub-pixel form (conv_up2.hip) against torch's fp32 ``F.conv2d(F.interpolate(x, 2))`` and its autograd.
Imported by tests/test_gpu_up2.py (benched shapes, default dispatch) and run by it as a child process with
MAS_CONV_WIDE_MIN_TILES_PER_CU=0 MAS_CONV_WIDE_ANY_WIDTH=1 (read once per process) so that SMALL shapes take the kernel:
    up2_check.py          ragged tiles, several cout tiles, K loops of 2 ... 16 stages, an XCD-rounded tile grid with empty tiles
    up2_check.py multi    with MAS_CONV_WGS_PER_CU=1: more tiles than work-groups (the persistent next-tile path of the benched launches)
Prints one line per case; exits non-zero on a mismatch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def _cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def _rel(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max() / b.detach().abs().max())


def run_case(case, dev, sample=None, tol=1e-2, check_stats=True):
    from mas_hip import ops
    ops.set_compute_dtype(torch.bfloat16)
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(cin + 3 * h + w)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).bfloat16().float()
    b = 0.1 * torch.randn(cout, generator=g)
    dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g).bfloat16()
    sample = sample or sorted({0, n // 2, n - 1})
    xs = x[sample].float().requires_grad_(True)
    ws, bs = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xs, scale_factor=2.0, mode="nearest"), ws, bs, padding=1)
    ref.backward(dy[sample].float())
    xd = _cl(x, dev).requires_grad_(True)
    wd, bd = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
    seen = []
    ops.set_launch_hook(lambda kind, shape, launch: (launch(), seen.append((kind, ops.last_kernel()))))
    try:
        y = ops.norm_act_conv(xd, wd, bd, stride=1, padding=(1, 1, 1, 1), upsample=True)
        dyd = torch.zeros_like(y)
        dyd[sample] = _cl(dy[sample], dev)
        y.backward(dyd)
    finally:
        ops.set_launch_hook(None)
    torch.cuda.synchronize()
    kinds = dict(seen)
    assert [k for kind, k in seen if kind == "conv_fwd"][0] == "conv_up2_fwd", seen       # (a later conv_fwd entry = the old data-gradient path)
    dfw = ops._desc(n, h, w, cin, 2 * h, 2 * w, cout, 3, 1, 1, 1, torch.bfloat16, torch.bfloat16, 0, True)
    if ops._up2_supported(dfw, True):                  # (128-channel output tiles, at least one tile per CU: else the 3x3 kernels at the high resolution)
        assert kinds.get("conv_up2_dgrad") == "conv_up2_dgrad", seen
    if ops._up2_wgrad_splits(dfw) > 0:                 # the weight gradient in the phase form too (conv_wgrad_dma.hip KS = 2 + mas_wgrad_reduce_up2)
        assert kinds.get("conv_wgrad") == "conv_wgrad_up2", seen
    e_y, e_x = _rel(y[sample], ref), _rel(xd.grad[sample], xs.grad)
    e_w, e_b = _rel(wd.grad, ws.grad), _rel(bd.grad, bs.grad)
    print(case, "fwd %.2e dgrad %.2e wgrad %.2e dbias %.2e" % (e_y, e_x, e_w, e_b))
    assert y.shape == (n, cout, 2 * h, 2 * w)
    assert e_y < tol and e_x < 1.5 * tol and e_w < tol and e_b < tol, (e_y, e_x, e_w, e_b)
    others = [i for i in range(n) if i not in sample]
    if others:                                         # images whose dy was zero get a zero gradient, and every image its output
        assert float(xd.grad[others].abs().max()) == 0.0
        assert torch.isfinite(y.detach()[others].float()).all() and float(y.detach()[others].float().abs().max()) > 0
    if check_stats:
        # the statistics the epilogue left for the consuming GroupNorm against the pass over the tensor
        part, rows = ops._take_stats(y)
        assert part is not None and rows == ((h + 15) // 16) * ((w + 31) // 32) * 4
        gam, bet = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        mr_f, ss_f = ops.gn_stats(y.detach(), gam, bet, 32, 1e-6, part, rows)
        mr_p, ss_p = ops.gn_stats(y.detach(), gam, bet, 32, 1e-6)
        assert _rel(mr_f[..., 1], mr_p[..., 1]) < 5e-3 and float((mr_f[..., 0] - mr_p[..., 0]).abs().max()) < 5e-3
    return y, xd.grad


CASES = [
    # n, cin, cout, h, w
    (2, 128, 128, 32, 32),
    (3, 64, 128, 24, 40),        # ragged rows and columns, the shortest K loop (2 stages forward)
    (2, 256, 256, 16, 32),       # two cout tiles, 8 forward stages
    (9, 128, 128, 16, 32),       # 9 spatial tiles: the XCD-rounded tile grid has 7 empty ones
    (1, 512, 512, 20, 33),       # four cout tiles, 16 stages, ragged
    (5, 128, 256, 17, 70),       # input and output channel counts differ both ways
    (4, 256, 128, 48, 31),       # a map narrower than a tile row
]
# more tiles than work-groups at one work-group per CU (256): every work-group walks 2-5 tiles, cout tile / phase / image change mid-walk
MULTI_CASES = [
    (20, 128, 128, 32, 64),      # 80 spatial tiles x 4 phases = 320 tiles
    (6, 256, 256, 40, 72),       # 54 -> 56 x 8 = 448 tiles, ragged, empty tail tiles
    (3, 512, 256, 16, 96),       # 9 -> 16 x 8
    (40, 64, 128, 16, 32),       # 40 x 4 = 160 forward tiles, but 40 data-gradient tiles: mixed
]

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    multi = len(sys.argv) > 1 and sys.argv[1] == "multi"
    if multi:
        print("multi-tile mode")
    bad = 0
    for case in (MULTI_CASES if multi else CASES):
        try:
            run_case(case, dev)
            print("ok   ", case)
        except AssertionError as e:
            bad += 1
            print("FAIL ", case, str(e)[:300])
    sys.exit(1 if bad else 0)
