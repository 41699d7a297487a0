"""The 8-row tiles of the stream convolution kernel (conv3x3_stream.hip, round 4: small maps fill the chip with twice the tiles):
bit-identical to the 16-row tiling of the same kernel, and both against F.conv2d in fp32 on the CPU; the data gradient of the
16x16 level through the autograd node the models use."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, th8):
    out = str(tmp_path / f"th8_{th8}.pt")
    env = dict(os.environ, MAS_CONV_STREAM_TH8=str(th8), PYTHONPATH=os.path.join(ROOT, "make-a-scene_amd"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "stream_th8_out.py"), out], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return torch.load(out)


def test_eight_row_tiles_equal_sixteen_row_tiles_and_the_cpu_reference(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from th8_spec import CASES, make_case
    y8, y16 = _run(tmp_path, 1), _run(tmp_path, 0)
    for case in CASES:
        name, n, cin, cout, h, w, res, ups = case
        assert torch.equal(y8[name], y16[name]), f"{name}: the two tilings differ"
        x, wt, b, r = make_case(case)
        sample = sorted({0, n // 2, n - 1})
        xs = x[sample].float()
        if ups:
            xs = F.interpolate(xs, scale_factor=2.0, mode="nearest")
        ref = F.conv2d(xs, wt, b, padding=1)
        if r is not None:
            ref = ref + r[sample].float()
        got = y8[name][sample].float()
        err = float((got - ref).abs().max() / ref.abs().max())
        print(f"{name}: max err / max |ref| = {err:.3e}")
        assert err < 1e-2, (name, err)


def test_level16_data_gradient_takes_the_stream_kernel_and_matches_fp32():
    """dL/dx of 512 -> 512 @16x16 at batch 32 (the transposed-weight forward of the same kernel, 8-row tiles) vs autograd in fp32"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    from mas_hip import ops
    dev = torch.device("cuda:0")
    ops.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    n, c, h = 32, 512, 16
    x = torch.randn(n, c, h, h, generator=g).bfloat16()
    w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).bfloat16().float()
    dy = torch.randn(n, c, h, h, generator=g).bfloat16()
    sample = [0, 17, 31]
    xs = x[sample].float().requires_grad_(True)
    F.conv2d(xs, w, None, padding=1).backward(dy[sample].float())
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(dev))
    seen = []
    ops.set_launch_hook(lambda kind, shape, launch: (launch(), seen.append((kind, ops.last_kernel()))))
    try:
        y = ops.norm_act_conv(xd, wd, None, stride=1, padding=(1, 1, 1, 1))
        y.backward(dy.to(dev).contiguous(memory_format=torch.channels_last))
    finally:
        ops.set_launch_hook(None)
    fwd = [k for kind, k in seen if kind == "conv_fwd"]
    assert fwd and all(k == "conv3x3_stream" for k in fwd), seen
    err = float((xd.grad[sample].float().cpu() - xs.grad).abs().max() / xs.grad.abs().max())
    print(f"dgrad 512 @16x16 N=32: {err:.3e}")
    assert err < 1.5e-2
