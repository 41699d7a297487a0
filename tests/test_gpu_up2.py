"""`Upsample` + 3x3 convolution in its sub-pixel form (conv_up2.hip, MAS_WLAYOUT_UP2; reference models/modules.py:44-59): four 2x2
phase convolutions over the LOW-resolution map must equal ``F.conv2d`` over the nearest-x2 image, and the data gradient straight
from dy (four phase images through the transposed phase weights) must equal autograd's -- whole and ragged tiles, several cout
tiles, K loops of 2 ... 16 stages, the benched shapes, fused GroupNorm statistics, bitwise run to run.  The packed phase image is
checked byte for byte against a numpy restatement of the layout."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def _rel(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max() / b.detach().abs().max())


R = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}          # 3x3 taps behind window position r of phase a


def _phase_image_reference(w, transpose):
    """numpy restatement of MAS_WLAYOUT_UP2 (include/mas_hip.h): [phase][chunk32][tap][row][64 B] as uint16 bf16 bit patterns"""
    cout, cin = w.shape[:2]
    rows, cols = (cin, cout) if transpose else (cout, cin)
    rows_pad, n_chunks = (rows + 127) // 128 * 128, (cols + 31) // 32
    wp = torch.zeros(2, 2, 2, 2, cout, cin)
    for a in (0, 1):
        for b in (0, 1):
            for r in (0, 1):
                for s in (0, 1):
                    acc = torch.zeros(cout, cin)
                    for kh in R[(a, r)]:
                        for kw in R[(b, s)]:
                            acc = acc + w[:, :, kh, kw]
                    wp[a, b, r, s] = acc
    out = np.zeros((4, n_chunks, 4, rows_pad, 32), dtype=np.uint16)
    for ph in range(4):
        a, b = ph >> 1, ph & 1
        for t in range(4):
            tr, ts = t >> 1, t & 1
            m = wp[a, b, 1 - tr, 1 - ts].t() if transpose else wp[a, b, tr, ts]      # [rows][cols]
            mb = torch.zeros(rows_pad, n_chunks * 32)
            mb[:rows, :cols] = m
            bits = mb.bfloat16().view(torch.int16).numpy().view(np.uint16)
            for row in range(rows_pad):
                frow = (row & ~127) + 4 * (row & 31) + ((row & 127) >> 5)
                for ch in range(n_chunks):
                    for sp in range(4):
                        ls = sp ^ ((row >> 2) & 3)
                        out[ph, ch, t, row, sp * 8:sp * 8 + 8] = bits[frow, ch * 32 + ls * 8: ch * 32 + ls * 8 + 8]
    return out.reshape(-1)


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("shape", [(128, 64), (256, 128), (128, 192)])
def test_packed_phase_image_bytes(shape, transpose):
    from mas_hip import ops, WLAYOUT_UP2
    dev = _dev()
    cout, cin = shape
    g = torch.Generator().manual_seed(cout + cin)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    got = ops.pack_conv_weight(w.to(dev), transpose, torch.bfloat16, WLAYOUT_UP2)
    ref = _phase_image_reference(w, transpose)
    got_bits = got.view(torch.int16).cpu().numpy().view(np.uint16)[:ref.size]
    assert np.array_equal(got_bits, ref), int((got_bits != ref).sum())


sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))


def _child(mode, extra_env):
    env = dict(os.environ, MAS_CONV_WIDE_MIN_TILES_PER_CU="0", MAS_CONV_WIDE_ANY_WIDTH="1", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "up2_check.py")] + mode, env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_small_shapes_vs_cpu_fp32():
    _dev()
    assert _child([], {}).count("ok   ") >= 7


def test_persistent_multi_tile_walk_vs_cpu_fp32():
    _dev()
    out = _child(["multi"], {"MAS_CONV_WGS_PER_CU": "1"})
    assert out.count("ok   ") >= 4 and "multi-tile mode" in out


@pytest.mark.parametrize("case", [(16, 128, 128, 128, 128), (16, 256, 256, 64, 64), (32, 512, 512, 32, 32)], ids=lambda c: "x".join(map(str, c)))
def test_benched_shapes(case):
    """the three Upsample layers of VQ-IMG that take the kernel (decoder.model[14,18,22]) at N = 16 (N = 32 at 512
    channels: its data gradient has one 128-channel tile per CU only at the benched batch); both kernels asserted"""
    from mas_hip import ops
    from up2_check import run_case
    n, cin, cout, h, w = case
    assert ops._up2_supported(ops._desc(n, h, w, cin, 2 * h, 2 * w, cout, 3, 1, 1, 1, torch.bfloat16, torch.bfloat16, 0, True), True)
    run_case(case, _dev(), sample=[0, n // 2 - 1, n - 1])


def test_bitwise_reproducible_and_follows_the_optimizer():
    from mas_hip import ops
    from mas_hip.optim import Adam
    dev = _dev()
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(2)
    x = _cl(torch.randn(32, 128, 32, 32).bfloat16(), dev).requires_grad_(True)       # 256 tiles: takes the kernel by default
    w = torch.nn.Parameter((torch.randn(128, 128, 3, 3) / 34.0).to(dev))
    opt = Adam([w], lr=0.05)
    outs = []
    for _ in range(2):
        x.grad = None
        y = ops.norm_act_conv(x, w, None, stride=1, padding=(1, 1, 1, 1), upsample=True)
        assert ops.last_kernel() == "conv_up2_fwd"
        y.float().square().mean().backward()
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
        w.grad = None
    assert all(torch.equal(u, v) for u, v in zip(outs[0], outs[1]))          # forward, data gradient and weight gradient: bitwise run to run
    y0 = outs[0][0]
    ops.norm_act_conv(x, w, None, stride=1, padding=(1, 1, 1, 1), upsample=True).float().square().mean().backward()
    opt.step()
    y1 = ops.norm_act_conv(x, w, None, stride=1, padding=(1, 1, 1, 1), upsample=True)
    ref = F.conv2d(F.interpolate(x.detach().float().cpu(), scale_factor=2.0, mode="nearest"), w.detach().cpu().bfloat16().float(), None, padding=1)
    assert _rel(y1, ref) < 1e-2 and _rel(y0, ref) > 5e-2          # the step moved the weights and both phase images followed
    x.grad = None
    y1.float().square().mean().backward()
    xr = x.detach().float().cpu().requires_grad_(True)
    F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), w.detach().cpu().bfloat16().float(), None, padding=1).square().mean().backward()
    assert _rel(x.grad, xr.grad) < 2e-2


def test_small_maps_keep_the_folded_kernels():
    """16x16 -> 32x32 (decoder.model[10]) is narrower than a tile row: the x2 stays folded into the 3x3 kernels' address arithmetic"""
    from mas_hip import ops
    dev = _dev()
    ops.set_compute_dtype(torch.bfloat16)
    x = _cl(torch.randn(4, 128, 16, 16).bfloat16(), dev)
    w = torch.nn.Parameter((torch.randn(128, 128, 3, 3) / 34.0).to(dev))
    y = ops.norm_act_conv(x, w, None, stride=1, padding=(1, 1, 1, 1), upsample=True)
    assert ops.last_kernel() != "conv_up2_fwd"
    ref = F.conv2d(F.interpolate(x.float().cpu(), scale_factor=2.0, mode="nearest"), w.detach().cpu().bfloat16().float(), None, padding=1)
    assert _rel(y, ref) < 1e-2


def test_large_batch_slices_stay_on_the_sub_pixel_kernels():
    """N = 160 at the last Upsample of VQ-IMG (128 -> 128, 128^2 -> 256^2): the output / dy are 2.68 GB, beyond the 31-bit byte offsets of
    the kernels' buffer descriptors.  ``ops`` cuts the forward and the data gradient into two batch slices that each take conv_up2
    (asserted), the weight gradient runs the phase-form kernel once per slice (asserted) and adds the slices' gradients in order.
    Forward and data gradient vs fp32 on images of both slices; the weight gradient vs the sum of its halves computed as separate
    launches (linearity in the batch: bitwise the same sums) and vs the 3x3 form of the same gradient (MAS-independent check: the
    folded kernels' weight gradient of a 16-image sample of both slices, compared through torch's fp32 autograd)."""
    from mas_hip import ops
    dev = _dev()
    ops.set_compute_dtype(torch.bfloat16)
    bf = torch.bfloat16
    n, c, h = 160, 128, 128
    assert len(ops._batch_slices(n, h * h * c * 2, 4 * h * h * c * 2)) == 2
    g = torch.Generator(device=dev).manual_seed(31)
    x = torch.randn(n, c, h, h, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device=dev, generator=g) / (9 * c) ** 0.5)
    seen = []
    ops.set_launch_hook(lambda kind, shape, launch: (launch(), seen.append((kind, ops.last_kernel()))))
    try:
        y = ops.norm_act_conv(x, w, None, stride=1, padding=(1, 1, 1, 1), upsample=True)
        dy = torch.randn(y.shape, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last)
        y.backward(dy)
        geo = (h, h, c, 2 * h, 2 * h, c, 3, 1, 1, 1)
        xa, xb = x.detach()[:80], x.detach()[80:]
        dwa, _ = ops.conv_wgrad_raw(xa, None, dy[:80], 80, *geo, 0, True, False)
        dwb, _ = ops.conv_wgrad_raw(xb, None, dy[80:], 80, *geo, 0, True, False)
    finally:
        ops.set_launch_hook(None)
    torch.cuda.synchronize()
    assert seen[0] == ("conv_fwd", "conv_up2_fwd") and ("conv_up2_dgrad", "conv_up2_dgrad") in seen, seen
    wg = [k for kind, k in seen if kind == "conv_wgrad"]
    assert wg == ["conv_wgrad_up2"] * 4, wg
    sample = [0, 79, 80, 159]
    xs = x.detach()[sample].float().cpu().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xs, scale_factor=2.0, mode="nearest"), w.detach().bfloat16().float().cpu(), None, padding=1)
    ref.backward(dy[sample].float().cpu())
    e_y, e_x = _rel(y[sample], ref), _rel(x.grad[sample], xs.grad)
    assert torch.equal(w.grad, dwa + dwb)                  # same launches, same order of addition
    ws = w.detach().bfloat16().float().cpu().requires_grad_(True)
    F.conv2d(F.interpolate(x.detach()[sample].float().cpu(), scale_factor=2.0, mode="nearest"), ws, None, padding=1).backward(dy[sample].float().cpu())
    dws, _ = ops.conv_wgrad_raw(x.detach()[sample].contiguous(memory_format=torch.channels_last), None, dy[sample].contiguous(memory_format=torch.channels_last),
                                len(sample), h, h, c, 2 * h, 2 * h, c, 3, 1, 1, 1, 0, True, False)
    e_w = _rel(dws, ws.grad)
    print("N=160 Upsample conv: forward %.3e, data gradient %.3e; weight gradient of the 4 sampled images vs fp32 autograd %.3e" % (e_y, e_x, e_w))
    assert e_y < 1e-2 and e_x < 1.5e-2 and e_w < 1e-3
