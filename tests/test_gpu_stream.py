"""conv3x3_stream.hip (the stream-scheduled 3x3 / stride-1 / bf16 kernel that carries the FLOPs) against a CPU fp32
convolution at SMALL shapes: ragged tiles, several pairs / cout tiles, upsample fold, pad 2, residual, GroupNorm(+SiLU)
prologue -- in both epilogue modes (deferred stores = default, immediate stores) --, and the same with more tiles than
work-groups (`test_stream_kernel_persistent_multi_tile_walk_vs_cpu`).  The real-shape check of the same kernel is
tests/test_gpu_parity_r2.py::test_stream_conv_real_shape_k64_vs_cpu_fp32 (a K64 pre-packed image keeps the launch on this kernel)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["3", "1"])
def test_stream_kernel_small_shapes_vs_cpu(mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MAS_CONV_STREAM=mode, MAS_CONV_STREAM_MIN_TILES_PER_CU="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "stream_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   ") >= 12


@pytest.mark.parametrize("mode", ["3", "1"])
def test_stream_kernel_persistent_multi_tile_walk_vs_cpu(mode):
    """one work-group per CU, 288-480 tiles per launch: every work-group walks 2 tiles (asserted from the launch geometry)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MAS_CONV_STREAM=mode, MAS_CONV_STREAM_MIN_TILES_PER_CU="0", MAS_CONV_WGS_PER_CU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "stream_check.py"), "multi"], env=env,
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   ") >= 6 and "multi-tile mode" in r.stdout


def test_stream_kernel_is_the_one_that_runs():
    """the dispatch really reaches the stream kernel for a qualifying shape: its result differs from the general kernel's only by
    accumulation order, and MAS_CONV_STREAM=0 (general kernel) agrees with it"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = ("import sys,os; sys.path.insert(0, os.path.join(%r, 'make-a-scene_amd')); import torch; from mas_hip import ops;"
            "g=torch.Generator().manual_seed(0); x=torch.randn(2,128,32,32,generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last);"
            "w=(torch.randn(128,128,3,3,generator=g)*0.03).cuda(); wp=ops.pack_conv_weight(w,False,torch.bfloat16);"
            "y=ops.conv_fwd_raw(x,None,wp,None,None,2,32,32,128,32,32,128,3,1,1,1,0,False,torch.bfloat16); torch.save(y.cpu(), sys.argv[1])") % ROOT
    import tempfile
    outs = []
    for mode in ("3", "0"):
        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as f:
            path = f.name
        env = dict(os.environ, MAS_CONV_STREAM=mode, MAS_CONV_STREAM_MIN_TILES_PER_CU="0")
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
        os.unlink(path)
    a, b = outs[0].float(), outs[1].float()
    assert float((a - b).abs().max() / b.abs().max()) < 1e-2
