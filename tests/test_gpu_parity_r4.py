"""Round-4 parity depth (VERDICT r3 "What's weak" 1-3).

* the decoder BACKWARD with the reference's own dL/drec injected and the reference's z_q fed in (tests/golden/vq_img256_dec_bwd.npz,
  made by tests/golden/make_golden_r4.py from the reference itself): fp32 and bf16, B=1 and the image replicated 16x -- the
  multi-tile launches bench.py times: 7 of the 11 dominant-shape weight gradients, every Upsample-fold weight gradient, conv_out's
  data gradient on conv_thin_fwd_kernel at 256^2 (reference models/modules.py:337-369, models/vqvae.py:26-29);
* the round-3 kernels at THEIR benched shapes with N=16, each against F.conv2d / autograd in fp32 on the CPU, with the kernel that
  ran asserted through ``mas_last_kernel``: Downsample forward / data gradient / weight gradient (conv_s2.hip), the RGB-edge layers
  (conv_thin.hip), the 1x1 GEMMs (conv1x1.hip);
* the fused GroupNorm statistics of the wide kernel's epilogue at the benched batch itself (32 x 128 ch x 256^2: 4096 tiles on the
  1024-work-group grid, four per work-group) and with one work-group per CU, against the stand-alone pass."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from test_gpu_parity_r3 import IMG, _build, _dev, rel_l2, relerr  # noqa: E402


# --------------------------------------------------------------------------------------------------------------
# 1. decoder backward under the reference's dL/drec
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,copies", [("fp32", 1), ("bf16", 1), ("bf16", 16)])
def test_img256_decoder_backward_with_reference_drec(golden_dir, mode, copies):
    """Our decoder is fed the REFERENCE's z_q and its backward the REFERENCE's dL/drec: what differs from the reference's decoder
    gradients is kernel / storage error only.  fp32 (exact-fp32 MFMA kernels): every recorded gradient and dL/dz_q within 2e-4
    rel-L2 / 5e-4 max-rel.  bf16: within 1.2x (rel-L2; 1.5x until round 5: measured 0.80-1.02x) / 2x (max-rel) of what the reference ITSELF loses under
    torch.autocast(bfloat16) on the CPU (recorded in the fixture), floors 3e-2 / 5e-2: the reference's autocast keeps GroupNorm and
    the residual stream in fp32, ours stores every tensor between kernels in bf16.  copies=16 replicates the image (every parameter
    gradient is 16x the reference's): the 256^2 / 128^2 layers then run the wide / LDS-DMA kernels with several tiles per work-group."""
    sys.path.insert(0, golden_dir)
    from r4_spec import DEC_GRADS
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "vq_img256_dec_bwd.npz"))
    g0 = np.load(os.path.join(golden_dir, "vq_img256.npz"))
    m = _build(IMG, 1, torch.float32 if mode == "fp32" else torch.bfloat16)
    zq = torch.from_numpy(g0["z_q"]).repeat(copies, 1, 1, 1).to(dev).requires_grad_(True)
    rec = m.decode(zq)
    rec.backward(torch.from_numpy(g["drec"]).repeat(copies, 1, 1, 1).to(dev))
    torch.cuda.synchronize()
    e_rec = relerr(rec[:1, :, ::8, ::8], g["rec_sub"])
    print("decoder forward on the reference z_q (%s, copies=%d): max-rel %.3e" % (mode, copies, e_rec))
    assert e_rec < (2e-3 if mode == "fp32" else 5e-2)
    params = dict(m.named_parameters())
    bad = []
    items = [(k, params[k].grad.detach().float().cpu()[sl] / copies, g["grad:" + k]) for k, sl in DEC_GRADS.items()]
    items.append(("dzq", zq.grad.detach().float().cpu()[:1], g["dzq"]))
    if copies > 1:
        assert torch.equal(zq.grad[0], zq.grad[copies - 1])                  # replicas are independent and deterministic
    for k, got, ref in items:
        e2, em = rel_l2(got, ref), relerr(got, ref)
        r2, rm = float(g["refbf16_l2:" + k]), float(g["refbf16_max:" + k])
        lim2, limm = (2e-4, 5e-4) if mode == "fp32" else (max(1.2 * r2, 3e-2), max(2.0 * rm, 5e-2))
        print("  %s copies=%d %-40s rel-L2 %.3e max-rel %.3e   (reference's own bf16 autocast: %.3e / %.3e)" % (mode, copies, k, e2, em, r2, rm))
        if e2 > lim2 or em > limm:
            bad.append((k, e2, em, lim2, limm))
    dec = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for n_, p in m.named_parameters()
                      if (n_.startswith("decoder.") or n_.startswith("post_quant_conv.")) and p.grad is not None)) / copies
    print("decoder backward under the reference drec (%s, copies=%d): gradient norm %.5f vs %.5f" % (mode, copies, dec, float(g["gradnorm_decoder"])))
    assert not bad, bad
    assert abs(dec - float(g["gradnorm_decoder"])) < (2e-4 if mode == "fp32" else 2e-2) * float(g["gradnorm_decoder"])


# --------------------------------------------------------------------------------------------------------------
# 2. round-3 kernels at their benched shapes, N = 16
# --------------------------------------------------------------------------------------------------------------
def _cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def _conv_case(n, cin, cout, h, ks, stride, pad4, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, h, generator=g).bfloat16()
    w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).bfloat16().float()
    b = 0.1 * torch.randn(cout, generator=g)
    t, bo, l, r = pad4
    ho = (h + t + bo - ks) // stride + 1
    dy = torch.randn(n, cout, ho, ho, generator=g).bfloat16()
    return x, w, b, dy, ho


def _reference(x, w, b, dy, stride, pad4, sample):
    """fp32 on the CPU: forward and data gradient on the sampled images only (images are independent), weight / bias gradient over
    the whole batch"""
    t, bo, l, r = pad4
    xs = x[sample].float().requires_grad_(True)
    ws = w.clone().requires_grad_(True)
    ys = F.conv2d(F.pad(xs, (l, r, t, bo)), ws, b, stride=stride)
    ys.backward(dy[sample].float())
    dx_s = xs.grad
    wf = w.clone().requires_grad_(True)
    bf_ = b.clone().requires_grad_(True)
    yf = F.conv2d(F.pad(x.float(), (l, r, t, bo)), wf, bf_, stride=stride)
    yf.backward(dy.float())
    return ys.detach(), dx_s, wf.grad, bf_.grad


@pytest.mark.parametrize("case", [
    # name, n, cin, cout, h, ks, stride, pad4, kernels expected (forward, data gradient, weight gradient)
    ("downsample_128_256", 16, 128, 128, 256, 3, 2, (0, 1, 0, 1), ("conv_s2_fwd", "conv_s2_dgrad", "wgrad_s2")),
    ("downsample_256_64", 16, 256, 256, 64, 3, 2, (0, 1, 0, 1), ("conv_s2_fwd", "conv_s2_dgrad", "wgrad_s2")),
    ("conv_in_rgb_256", 16, 3, 128, 256, 3, 1, (1, 1, 1, 1), ("conv_thin_fwd", None, "wgrad_thin")),
    ("conv_out_rgb_256", 16, 128, 3, 256, 3, 1, (1, 1, 1, 1), ("conv_thin_out", "conv_thin_fwd", "wgrad_thin")),     # (forward: round 5)
    ("nin_shortcut_256_128_128", 16, 256, 128, 128, 1, 1, (0, 0, 0, 0), ("conv1x1", "conv1x1", "wgrad1x1")),
    ("nin_shortcut_512_256_64", 16, 512, 256, 64, 1, 1, (0, 0, 0, 0), ("conv1x1", "conv1x1", "wgrad1x1")),
], ids=lambda c: c[0])
def test_round3_kernels_at_their_benched_shapes_vs_cpu_fp32(case):
    """through the autograd node the models use (ops.norm_act_conv): forward, data gradient and weight / bias gradient of one layer
    at the shape bench.py runs it, N=16, bf16, against F.conv2d + autograd in fp32 on the CPU (forward / data gradient on three
    sampled images, parameter gradients over the whole batch).  Tolerance 1e-2 of the tensor's maximum (bf16 operands are shared with
    the reference; the results are stored in bf16, the parameter gradients in fp32)."""
    from mas_hip import ops
    dev = _dev()
    name, n, cin, cout, h, ks, stride, pad4, want = case
    ops.set_compute_dtype(torch.bfloat16)
    x, w, b, dy, ho = _conv_case(n, cin, cout, h, ks, stride, pad4, seed=len(name) + cin)
    sample = [0, n // 2, n - 1]
    y_ref, dx_ref, dw_ref, db_ref = _reference(x, w, b, dy, stride, pad4, sample)
    xd = _cl(x, dev).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(dev))
    bd = torch.nn.Parameter(b.to(dev))
    seen = []

    def hook(kind, shape, launch):
        launch()
        seen.append((kind, ops.last_kernel()))

    ops.set_launch_hook(hook)
    try:
        y = ops.norm_act_conv(xd, wd, bd, stride=stride, padding=pad4)
        k_fwd = ops.last_kernel()
        y.backward(_cl(dy, dev))
        k_last = ops.last_kernel()                               # the data gradient is the backward's last launch
    finally:
        ops.set_launch_hook(None)
    torch.cuda.synchronize()
    kinds = dict(fwd=k_fwd, dgrad=k_last)
    for kind, kern in seen[1:]:
        if kind == "conv_wgrad":
            kinds.setdefault("wgrad", kern)
    print(name, "kernels:", kinds)
    if want[0]:
        assert kinds["fwd"] == want[0], kinds
    if want[1]:
        assert kinds["dgrad"] == want[1], kinds
    if want[2]:
        assert kinds["wgrad"] == want[2], kinds
    e_y = relerr(y[sample], y_ref)
    e_dx = relerr(xd.grad[sample], dx_ref)
    e_dw, e_db = relerr(wd.grad, dw_ref), relerr(bd.grad, db_ref)
    print("%s N=%d: fwd %.3e dgrad %.3e wgrad %.3e (rel-L2 %.3e) bias %.3e" % (name, n, e_y, e_dx, e_dw, rel_l2(wd.grad, dw_ref), e_db))
    assert e_y < 1e-2 and e_dx < 1.5e-2 and e_dw < 1e-2 and e_db < 1e-2


# --------------------------------------------------------------------------------------------------------------
# 3. fused GroupNorm statistics, several tiles per work-group
# --------------------------------------------------------------------------------------------------------------
def test_fused_statistics_at_the_benched_batch_equal_the_standalone_pass():
    """32 x 128 ch x 256^2, the launch bench.py times 11 + 10 times per step: 4096 tiles on 1024 work-groups (asserted), the statistics
    epilogue flushes once per finished tile while the next tile's weight DMA is in flight.  The table must give the same mean / rstd /
    scale / shift as mas_gn_stats' pass over the stored tensor, with and without the residual epilogue, bitwise run to run."""
    from mas_hip import ops
    dev = _dev()
    bf = torch.bfloat16
    n, c, h = 32, 128, 256
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(n, c, h, h, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, 3, 3, device=dev, generator=g) / (9 * c) ** 0.5
    b = 0.1 * torch.randn(c, device=dev, generator=g)
    r = torch.randn(n, c, h, h, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last)
    ops._stats_state["on"] = True
    tiles = n * (h // 16) * (h // 32)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    assert tiles == 4096 and tiles >= 4 * 4 * cus                            # >= 4 tiles per work-group of the 4-per-CU persistent grid
    ga, be = 1.0 + 0.1 * torch.randn(c, device=dev, generator=g), 0.1 * torch.randn(c, device=dev, generator=g)
    for res in (None, r):
        y, part, rows = ops.conv_fwd_raw(x, None, ops.ConvWeight(wt, False), b, res, n, h, h, c, h, h, c, 3, 1, 1, 1, 0, False, bf, want_stats=True)
        assert ops.last_kernel() == "conv3x3_wide" and part is not None and rows == (h // 16) * (h // 32)
        mr_f, ss_f = ops.gn_stats(y, ga, be, 32, 1e-6, part, rows)
        mr_s, ss_s = ops.gn_stats(y, ga, be, 32, 1e-6)
        assert float((mr_f - mr_s).abs().max() / mr_s.abs().max()) < 1e-4
        assert float((ss_f - ss_s).abs().max() / ss_s.abs().max()) < 1e-4
        y2, part2, _ = ops.conv_fwd_raw(x, None, ops.ConvWeight(wt, False), b, res, n, h, h, c, h, h, c, 3, 1, 1, 1, 0, False, bf, want_stats=True)
        assert torch.equal(part, part2) and torch.equal(y, y2)


def test_fused_statistics_with_one_work_group_per_cu():
    """MAS_CONV_WGS_PER_CU=1 (a subprocess: the knob is read once per process): 1024-4096 tiles on 256 work-groups, 4-16 tiles per
    work-group, ragged tiles and residual included -- the fused table against the stand-alone pass (tests/helpers/stats_check.py)."""
    _dev()
    env = dict(os.environ, MAS_CONV_WGS_PER_CU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "stats_check.py")], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   ") >= 4


# --------------------------------------------------------------------------------------------------------------
# 4. tensors of 2^31 bytes and more (per-GPU batches >= 128 at 128 ch @256^2: conf/img_config.yaml:17 hints at 192)
# --------------------------------------------------------------------------------------------------------------
def test_large_batch_stays_on_the_fast_kernels():
    """N = 160 at the dominant shape: x / y are 2.68 GB each, beyond the 31-bit byte offsets of the fast kernels' buffer descriptors.
    ops cuts such launches into batch slices below the limit (images are independent): the forward must run on conv3x3_wide and the
    weight gradient on conv_wgrad_dma (asserted), the forward must match F.conv2d fp32 on images sampled from both slices, the weight
    gradient must equal the sum of the two halves' weight gradients computed as ordinary (single-slice) launches -- linearity in the
    batch -- and the fused GroupNorm statistics must equal the stand-alone pass over the whole 2.68 GB tensor."""
    from mas_hip import ops
    dev = _dev()
    bf = torch.bfloat16
    n, c, h = 160, 128, 256
    assert len(ops._batch_slices(n, h * h * c * 2, h * h * c * 2)) == 2
    g = torch.Generator(device=dev).manual_seed(21)
    x = torch.randn(n, c, h, h, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device=dev, generator=g) / (9 * c) ** 0.5)
    b = 0.1 * torch.randn(c, device=dev, generator=g)
    geo = (h, h, c, h, h, c, 3, 1, 1, 1)
    seen = []
    ops.set_launch_hook(lambda kind, shape, launch: (launch(), seen.append((kind, ops.last_kernel()))))
    try:
        ops._stats_state["on"] = True
        y, part, rows = ops.conv_fwd_raw(x, None, ops.ConvWeight(w, False), b, None, n, *geo, 0, False, bf, want_stats=True)
        ga, be = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        mr_f, ss_f = ops.gn_stats(y, ga, be, 32, 1e-6, part, rows)
        mr_s, ss_s = ops.gn_stats(y, ga, be, 32, 1e-6)
        dy = torch.randn(n, c, h, h, device=dev, generator=g).to(bf).contiguous(memory_format=torch.channels_last)
        dw, db = ops.conv_wgrad_raw(x, None, dy, n, *geo, 0, False, True)
        dwa, dba = ops.conv_wgrad_raw(x[:80], None, dy[:80], 80, *geo, 0, False, True)
        dwb, dbb = ops.conv_wgrad_raw(x[80:], None, dy[80:], 80, *geo, 0, False, True)
    finally:
        ops.set_launch_hook(None)
    torch.cuda.synchronize()
    assert seen[0] == ("conv_fwd", "conv3x3_wide") and [k for k in seen if k[0] == "conv_wgrad"] == [("conv_wgrad", "conv_wgrad_dma")] * 3, seen
    assert part is not None and float((mr_f - mr_s).abs().max() / mr_s.abs().max()) < 1e-4 and float((ss_f - ss_s).abs().max() / ss_s.abs().max()) < 1e-4
    sample = [0, 79, 80, 159]
    ref = F.conv2d(x[sample].float().cpu(), w.detach().bfloat16().float().cpu(), b.cpu(), padding=1)
    e = relerr(y[sample], ref)
    print("N=160 wide forward on sampled images of both slices: %.3e" % e)
    assert e < 1e-2
    e_w = float((dw - (dwa + dwb)).abs().max() / dw.abs().max())
    e_b = float((db - (dba + dbb)).abs().max() / db.abs().max())
    print("N=160 weight gradient vs the sum of its two halves: %.3e, bias %.3e" % (e_w, e_b))
    assert e_w < 1e-5 and e_b < 1e-5
