"""The single-head spatial self-attention core of AttnBlock (reference models/modules.py:174-187) on the hand-written kernels
(``mas_spatial_attn_fwd / _bwd``) against torch fp32 on CPU: forward and the gradient w.r.t. the fused q|k|v
projection, at the reference's shapes (16x16x512 VQ-IMG, 8x8x512 VQ-SEG) and at ragged / tiny ones."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-12))


def relmax(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("case", [
    # n, c, h, w
    (2, 512, 16, 16),     # every AttnBlock of VQ-IMG (conf/img_config.yaml)
    (3, 512, 8, 8),       # VQ-SEG's latent grid
    (2, 256, 16, 16),
    (1, 128, 5, 7),       # 35 tokens: ragged second 32-token block
    (2, 32, 4, 4),        # fewer tokens than one block, one channel tile
    (1, 64, 16, 16),
    (32, 512, 16, 16),    # the benched launch: 256 work-groups, XCD-aware block order, every rotation of the tile walk
    (1, 96, 6, 6),        # 12 units per row (the unit -> (row, column) walk carries), channels padded 96 -> 128 in LDS
    (2, 160, 9, 9),       # two channel groups, the second one a quarter full; 81 tokens
    (1, 384, 16, 16),     # three channel groups: one ring slot per tile loads a clamped line and skips its MFMAs
])
def test_spatial_attention_fwd_bwd_vs_torch(case):
    from mas_hip import ops
    n, c, h, w = case
    dev = _dev()
    g = torch.Generator().manual_seed(c + h)
    qkv = (torch.randn(n, 3 * c, h, w, generator=g) * 1.5).bfloat16()
    go = torch.randn(n, c, h, w, generator=g).bfloat16()
    ref_in = qkv.float().requires_grad_(True)
    t = ref_in.permute(0, 2, 3, 1).reshape(n, h * w, 3 * c)
    q, k, v = t[..., :c], t[..., c:2 * c], t[..., 2 * c:]
    p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (c ** -0.5), dim=2)
    ref = torch.bmm(p, v).reshape(n, h, w, c).permute(0, 3, 1, 2)
    ref.backward(go.float())
    x = qkv.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    calls = []
    orig = ops._SpatialAttention.forward
    ops._SpatialAttention.forward = staticmethod(lambda ctx, a, b: (calls.append(1), orig(ctx, a, b))[1])
    try:
        y = ops.spatial_attention(x, c)
    finally:
        ops._SpatialAttention.forward = staticmethod(orig)
    assert calls, "the HIP kernel was not dispatched"
    y.backward(go.to(dev))
    assert y.shape == (n, c, h, w) and y.dtype == torch.bfloat16
    print(f"{case}: fwd rel-L2 {rel_l2(y, ref):.2e} max {relmax(y, ref):.2e}; dqkv rel-L2 {rel_l2(x.grad, ref_in.grad):.2e} max {relmax(x.grad, ref_in.grad):.2e}")
    assert relmax(y, ref) < 2e-2 and rel_l2(y, ref) < 1e-2
    assert rel_l2(x.grad, ref_in.grad) < 2e-2 and relmax(x.grad, ref_in.grad) < 4e-2
    # deterministic (no atomics): a second backward gives the same bits
    x2 = qkv.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ops.spatial_attention(x2, c).backward(go.to(dev))
    assert torch.equal(x2.grad, x.grad)


def test_fp32_mode_keeps_the_library_path():
    from mas_hip import ops
    dev = _dev()
    x = torch.randn(1, 96, 4, 4, device=dev).contiguous(memory_format=torch.channels_last)
    y = ops.spatial_attention(x, 32)
    assert y.dtype == torch.float32 and y.shape == (1, 32, 4, 4)
