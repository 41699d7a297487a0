"""``mas_pack_conv_weight_tiles`` (misc.hip: every bf16 image of a parameter from one LDS-tiled read of it) against the per-image gather
kernel ``mas_pack_conv_weight_layout``: the packed images must be bit-identical, zero padding included -- the convolution kernels copy
them linearly into LDS without bounds checks (include/mas_hip.h, "weight packing")."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# Cout, Cin, ks: the model's shapes, channel counts that are not multiples of the 64 x 64 tile / of a 16-byte run, 1x1 and 4x4 filters
SHAPES = [(128, 128, 3), (256, 128, 3), (128, 256, 3), (512, 512, 3), (512, 256, 1), (256, 256, 1), (1536, 512, 1), (96, 40, 3), (8, 128, 3),
          (128, 8, 3), (200, 72, 1), (64, 128, 4), (130, 66, 3), (3, 5, 3)]


@pytest.mark.parametrize("shape", SHAPES)
def test_tiled_pack_equals_the_gather_kernel(shape):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mas_hip
    from mas_hip import ops
    dev = torch.device("cuda:0")
    cout, cin, ks = shape
    g = torch.Generator().manual_seed(cout * 7 + cin + ks)
    w = torch.nn.Parameter(torch.randn(cout, cin, ks, ks, generator=g).to(dev))
    combos = [(False, mas_hip.WLAYOUT_K64), (True, mas_hip.WLAYOUT_K64)]
    if ks == 3:
        combos += [(False, mas_hip.WLAYOUT_K32), (True, mas_hip.WLAYOUT_K32)]
    cache = ops._PackCache()
    # poison the buffers the cache allocates (torch.empty): padding that the tiled kernel failed to write would show
    for tr, lay in combos:
        n = mas_hip.lib().mas_packed_weight_elems(cout, cin, ks)
        key = (id(w), tr, torch.bfloat16, lay)
        import weakref
        cache.store[key] = [weakref.ref(w), None, torch.full((n,), float("nan"), dtype=torch.bfloat16, device=dev)]
    got = {c: cache.get(w, c[0], torch.bfloat16, c[1]) for c in combos}       # the first get repacks all four in one launch
    assert ops.last_kernel() == "pack_conv_weight_tiles"
    torch.cuda.synchronize()
    for (tr, lay), img in got.items():
        ref = ops.pack_conv_weight(w.detach(), tr, torch.bfloat16, lay)
        assert img.shape == ref.shape
        # the image proper: [chunks of CK columns][taps][rows padded to 128][CK]; the buffer (mas_packed_weight_elems: sized for either
        # orientation) may be longer -- what lies behind the image is never read by a kernel
        rows, cols = (cin, cout) if tr else (cout, cin)
        ck = 32 if lay == mas_hip.WLAYOUT_K32 else 64
        used = ks * ks * -(-cols // ck) * (-(-rows // 128) * 128) * ck
        a, b = img.view(torch.int16)[:used], ref.view(torch.int16)[:used]
        assert torch.equal(a, b), (shape, tr, lay, int((a != b).sum()), used)
        assert not torch.isnan(img[:used].float()).any()           # all of it written (the buffer was poisoned)


def test_tiled_pack_follows_the_optimizer():
    """two parameters, an optimizer step: the stale images of both are rewritten by one launch and equal a fresh gather pack"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mas_hip
    from mas_hip import ops
    dev = torch.device("cuda:0")
    ws = [torch.nn.Parameter(torch.randn(128, 128, 3, 3, device=dev)), torch.nn.Parameter(torch.randn(256, 128, 1, 1, device=dev))]
    opt = torch.optim.Adam(ws, lr=1e-2, fused=True)
    cache = ops._pack_cache
    for _ in range(2):
        imgs = [cache.get(w, tr, torch.bfloat16, mas_hip.WLAYOUT_K64) for w in ws for tr in (False, True)]
        refs = [ops.pack_conv_weight(w.detach(), tr, torch.bfloat16, mas_hip.WLAYOUT_K64) for w in ws for tr in (False, True)]
        assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(imgs, refs))
        for w in ws:
            w.grad = torch.randn_like(w)
        opt.step()
