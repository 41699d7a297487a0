"""Seeded random sweep of the convolution dispatch (round 6): whatever kernel `mas_conv_fwd / _dgrad / _wgrad` picks for a shape -- wide, stream,
sub-pixel, stride-2, thin, 1x1 or the general kernels -- forward, data gradient, weight and bias gradient must match torch's fp32 convolution
(and GroupNorm+SiLU in front of it, and the residual behind it) on the same bf16-rounded operands.  Complements the hand-picked cases of
tests/test_gpu_wide.py / test_gpu_stream.py / test_gpu_up2.py / test_gpu_parity_r3.py: shapes nobody chose (odd sizes, ragged tiles, channel
counts that change the kernel class, several images), every one through ``ops.norm_act_conv`` exactly as the model calls it
(reference models/modules.py:49,68,93,100,107,113: every nn.Conv2d of the VQ model)."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cases():
    rnd = random.Random(20260)
    out = []
    for i in range(28):
        ks = rnd.choice([3, 3, 3, 1])
        stride = 2 if (ks == 3 and rnd.random() < 0.15) else 1
        ups = ks == 3 and stride == 1 and rnd.random() < 0.2
        cin = rnd.choice([8, 64, 128, 128, 192, 256, 320, 512])
        cout = rnd.choice([8, 64, 128, 128, 256, 384, 512])
        n = rnd.choice([1, 2, 3, 5])
        h, w = rnd.randint(5, 70), rnd.randint(5, 70)
        if cin * cout >= 512 * 256:
            h, w = min(h, 24), min(w, 40)
        act = 0 if (ks == 1 or stride == 2 or ups or cin % 32) else rnd.choice([0, 0, 2])
        res = stride == 1 and not ups and rnd.random() < 0.4
        out.append((i, n, cin, cout, h, w, ks, stride, ups, act, res))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "c%d_n%d_%dto%d_%dx%d_k%d_s%d%s%s%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], "_up" if c[8] else "",
                                                                                              "_gn" if c[9] else "", "_res" if c[10] else ""))
def test_random_conv_vs_torch_fp32(case):
    from mas_hip import ops
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    i, n, cin, cout, h, w, ks, stride, ups, act, res = case
    old = ops.compute_dtype()
    ops.set_compute_dtype(torch.bfloat16)
    try:
        g = torch.Generator().manual_seed(1000 + i)
        x = torch.randn(n, cin, h, w, generator=g).bfloat16()
        wt = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
        b = 0.1 * torch.randn(cout, generator=g)
        gw, gb = 1.0 + 0.2 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
        if stride == 2:
            pad4 = (0, 1, 0, 1)                      # the Downsample geometry (reference modules.py:76-78)
        else:
            pad4 = (ks // 2,) * 4
        # reference in fp32 on the CPU
        xr = x.float().requires_grad_(True)
        wr = wt.bfloat16().float().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        hr = xr
        gwr = gbr = None
        if act:
            gwr, gbr = gw.clone().requires_grad_(True), gb.clone().requires_grad_(True)
            hr = F.silu(F.group_norm(hr, 32, gwr, gbr, 1e-6))
        if ups:
            hr = F.interpolate(hr, scale_factor=2.0, mode="nearest")
        hr = F.pad(hr, (pad4[2], pad4[3], pad4[0], pad4[1]))
        yr = F.conv2d(hr, wr, br, stride=stride)
        rr = torch.randn(yr.shape, generator=g).bfloat16() if res else None
        if res:
            yr = yr + rr.float()
        dy = torch.randn(yr.shape, generator=g).bfloat16()
        yr.backward(dy.float())
        # ours
        xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
        gwd = torch.nn.Parameter(gw.to(dev)) if act else None
        gbd = torch.nn.Parameter(gb.to(dev)) if act else None
        rd = rr.to(dev).contiguous(memory_format=torch.channels_last) if res else None
        y = ops.norm_act_conv(xd, wd, bd, gwd, gbd, residual=rd, stride=stride, padding=pad4, act=act, upsample=ups)
        y.backward(dy.to(dev).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype(old)
    rel = lambda a, c: float((a.detach().float().cpu() - c.detach()).abs().max() / (c.detach().abs().max() + 1e-30))
    tol = 3e-2 if act else 1.5e-2                    # bf16 storage of the normalised activation in front of the convolution
    assert y.shape == yr.shape, (case, y.shape, yr.shape)
    errs = dict(y=rel(y, yr), dx=rel(xd.grad, xr.grad), dw=rel(wd.grad, wr.grad), db=rel(bd.grad, br.grad))
    if act:
        errs.update(dgn_w=rel(gwd.grad, gwr.grad), dgn_b=rel(gbd.grad, gbr.grad))
    assert all(v < tol for v in errs.values()), (case, errs)
