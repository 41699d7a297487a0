"""CPU restatements of the index algebra the round-3 kernels are built on, checked against torch's own convolution / autograd.
No GPU, no library: these pin the DECOMPOSITIONS (which taps a parity class owns, which way the thin tensor is shifted, how a 1x1
weight gradient is a pixel-contraction GEMM), so that a GPU parity failure can be told apart from a wrong derivation.

  * conv_s2.hip  conv_s2_dgrad_kernel: dx by parity class (u & 1, v & 1), taps kh in {0, 2} / {1}, dy row (u >> 1) - (kh >> 1)
  * conv_s2.hip  conv_s2_fwd_kernel / wgrad_s2_kernel: x pixel (2 i + kh, 2 j + kw), zero beyond the map (the one-sided padding)
  * conv_thin.hip wgrad_thin_kernel: D[(tap, cs)][cb] = sum_p S[p + sgn (tap - 1)][cs] * B[p][cb], sgn = -1 (conv_out) / +1 (conv_in)
  * conv1x1.hip  wgrad1x1_kernel + mas_wgrad_reduce: split-K over pixel chunks, slabs added in a fixed order
(reference sites: models/modules.py:62-81 Downsample, :219 conv_in, :345 conv_out, :106-108 / :145-160 the 1x1 layers)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _rand(*shape, seed=0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float64))


@pytest.mark.parametrize("h,w", [(8, 10), (9, 7), (12, 5)])
def test_stride2_data_gradient_by_parity_class(h, w):
    n, cin, cout = 2, 3, 4
    x = _rand(n, cin, h, w, seed=1).requires_grad_(True)
    wt = _rand(cout, cin, 3, 3, seed=2)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, stride=2)
    dy = _rand(*y.shape, seed=3)
    y.backward(dy)
    ho, wo = y.shape[2:]
    dx = torch.zeros_like(x)
    for p in (0, 1):
        for q in (0, 1):
            for kh in ((0, 2) if p == 0 else (1,)):
                for kw in ((0, 2) if q == 0 else (1,)):
                    for u in range(p, h, 2):
                        a = (u >> 1) - (kh >> 1)
                        if not 0 <= a < ho:
                            continue
                        for v in range(q, w, 2):
                            b = (v >> 1) - (kw >> 1)
                            if 0 <= b < wo:
                                dx[:, :, u, v] += torch.einsum("no,oi->ni", dy[:, :, a, b], wt[:, :, kh, kw])
    assert torch.allclose(dx, x.grad, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("h,w", [(8, 10), (9, 7)])
def test_stride2_forward_and_weight_gradient_addressing(h, w):
    n, cin, cout = 2, 3, 4
    x = _rand(n, cin, h, w, seed=4)
    wt = _rand(cout, cin, 3, 3, seed=5).requires_grad_(True)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, stride=2)
    dy = _rand(*y.shape, seed=6)
    y.backward(dy)
    ho, wo = y.shape[2:]

    def xat(i, j):                                    # zero beyond the map: the padded row / column is never materialised
        return x[:, :, i, j] if i < h and j < w else torch.zeros(n, cin, dtype=x.dtype)

    yy = torch.zeros_like(y)
    dw = torch.zeros_like(wt)
    for i in range(ho):
        for j in range(wo):
            for kh in range(3):
                for kw in range(3):
                    xv = xat(2 * i + kh, 2 * j + kw)
                    yy[:, :, i, j] += torch.einsum("ni,oi->no", xv, wt[:, :, kh, kw].detach())
                    dw[:, :, kh, kw] += torch.einsum("no,ni->oi", dy[:, :, i, j], xv)
    assert torch.allclose(yy, y.detach(), rtol=1e-10, atol=1e-10)
    assert torch.allclose(dw, wt.grad, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("big_is_x", [True, False])
def test_thin_weight_gradient_is_a_shifted_pixel_contraction(big_is_x):
    """conv_out (big = x with 128 -> here 6 channels, thin = dy) and conv_in (big = dy, thin = x): one formula, sgn = -1 / +1."""
    n, h, w, cb, cs = 2, 5, 6, 6, 2
    cin, cout = (cb, cs) if big_is_x else (cs, cb)
    x = _rand(n, cin, h, w, seed=7)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    dy = _rand(n, cout, h, w, seed=8)
    F.conv2d(x, wt, padding=1).backward(dy)
    big, thin, sgn = (x, dy, -1) if big_is_x else (dy, x, 1)
    d = torch.zeros(9, cs, cb, dtype=torch.float64)   # D[(tap, cs)][cb]
    for tap in range(9):
        kh, kw = divmod(tap, 3)
        for i in range(h):
            for j in range(w):
                si, sj = i + sgn * (kh - 1), j + sgn * (kw - 1)
                if 0 <= si < h and 0 <= sj < w:       # out-of-range halo pixels read as zeros
                    d[tap] += torch.einsum("ns,nb->sb", thin[:, :, si, sj], big[:, :, i, j])
    got = d.permute(1, 0, 2).reshape(cs, 3, 3, cb).permute(0, 3, 1, 2) if big_is_x else d.permute(2, 1, 0).reshape(cb, cs, 3, 3)
    assert torch.allclose(got, wt.grad, rtol=1e-10, atol=1e-10)


def test_pointwise_weight_gradient_split_k_slabs_in_fixed_order():
    """dW[co][ci] = sum_p dy[p][co] a[p][ci] over 64-pixel chunks dealt round-robin to nsplit work-groups; the slabs are added in slab
    order, so the result does not depend on which work-group finishes first -- and equals autograd's up to fp32 summation order."""
    m, cin, cout, nsplit = 1000, 8, 12, 5
    rng = np.random.default_rng(9)
    a = rng.standard_normal((m, cin)).astype(np.float32)
    dy = rng.standard_normal((m, cout)).astype(np.float32)
    n_chunks = (m + 63) // 64
    slabs = np.zeros((nsplit, cout, cin), np.float32)
    for c in range(n_chunks):
        sl = slice(64 * c, min(m, 64 * c + 64))
        slabs[c % nsplit] += dy[sl].T @ a[sl]
    dw = np.zeros((cout, cin), np.float32)
    for sp in range(nsplit):                          # fixed order
        dw = dw + slabs[sp]
    ref = dy.astype(np.float64).T @ a.astype(np.float64)
    assert np.abs(dw - ref).max() < 1e-3 * np.abs(ref).max()
    again = sum(slabs[sp] for sp in range(nsplit))    # the same order again: bitwise the same
    assert np.array_equal(dw, np.asarray(again, np.float32))


def test_batch_slices_keep_every_tensor_below_2_to_the_31():
    """ops._batch_slices (host arithmetic): launches whose tensors reach 2^31 bytes are cut into even batch slices below the limit the
    fast kernels' 31-bit buffer offsets impose; smaller launches are left alone"""
    from mas_hip import ops
    img = 256 * 256 * 128 * 2
    assert ops._batch_slices(32, img, img) == [(0, 32)]
    assert ops._batch_slices(127, img, img) == [(0, 127)]
    assert ops._batch_slices(128, img, img) == [(0, 64), (64, 128)]
    assert ops._batch_slices(160, img, img) == [(0, 80), (80, 160)]
    for n, a, b in ((192, img, img), (300, img, 2 * img), (7, 700_000_000, 1), (1, 3_000_000_000, 1), (1000, img // 4, img)):
        sl = ops._batch_slices(n, a, b)
        assert sl[0][0] == 0 and sl[-1][1] == n and all(p[1] == q[0] for p, q in zip(sl, sl[1:]))
        assert all((n1 - n0) * max(a, b) <= (1 << 31) - 1 or n1 - n0 == 1 for n0, n1 in sl)
        assert max(n1 - n0 for n0, n1 in sl) - min(n1 - n0 for n0, n1 in sl) <= max(1, len(sl))


@pytest.mark.parametrize("th", [16, 8])
def test_stream_kernel_staging_plan_covers_the_halo_patch_once(th):
    """conv3x3_stream.hip, SGeo<TH> + slot_pix: the (TH + 2) x 18-pixel halo patch is moved by 1-KiB DMA pieces of 8 pixels x 128 B;
    wave w issues pieces w, w + 8, ... and REPEATS its previous piece where w + 8 i runs past the last one (every wave must issue
    the same number of VMEM operations: the kernel's waits are counted).  Every live patch pixel must be written (>= once, by one
    piece), dead pixels of the last piece must be flagged out of range, and the per-stage schedule of the plain path (PCNT pieces in
    each of three stages) must add up to the slot count.  TH = 8 is round 4's tile for small maps."""
    npix = (th + 2) * 18
    npiece = (npix * 128 + 1023) // 1024
    nslot = (npiece + 7) // 8
    pcnt = nslot // 3
    assert (npiece, nslot, pcnt) == ((41, 6, 2) if th == 16 else (23, 3, 1))
    assert 3 * pcnt == nslot                                   # stages 0..2 (chunk B) and 5..7 (next chunk A) issue all slots
    written = np.zeros(npiece * 8, dtype=int)
    for wave in range(8):
        seen = []
        for i in range(nslot):
            piece = wave + 8 * i if wave + 8 * i < npiece else wave + 8 * (i - 1)
            assert 0 <= piece < npiece
            seen.append(piece)
            for lrow in range(8):
                q = piece * 8 + lrow
                pr = (q * 3641) >> 16                          # the kernel's q / 18
                assert pr == q // 18
                live = q < npix
                if live:
                    assert 0 <= pr < th + 2 and 0 <= q - 18 * pr < 18
                written[q] += 1
        assert len(set(seen)) >= nslot - 1                     # at most one repeated piece per wave
    assert (written[:npix] >= 1).all()                         # every live pixel arrives
    # a repeated piece rewrites the same bytes (same source address): benign; pieces are otherwise disjoint
    per_piece = written.reshape(npiece, 8)[:, 0]
    assert per_piece.min() >= 1 and per_piece.max() <= 2
    # fragment addressing: wave_p in 0..3 owns NJ = TH / 8 fragments of 32 pixels; together they tile TH x 16 output pixels
    nj = th // 8
    pix = sorted((wp * nj + j) * 32 + l for wp in range(4) for j in range(nj) for l in range(32))
    assert pix == list(range(th * 16))
    # deferred stores: 4 NJ per lane, one per stage, all within the 8 stages that follow the tile
    assert 4 * nj <= 8


# ---------------------------------------------------------------------------------------------------------------------------------
# conv_up2.hip (round 5): `Upsample` + 3x3 (reference models/modules.py:44-59) as four 2x2 phase convolutions on the low-resolution map
# ---------------------------------------------------------------------------------------------------------------------------------
_R = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}          # 3x3 taps behind window position r of phase a


def _phase_weights(wt):
    """Wp[a][b][r][s] = sum of W[kh][kw] over kh in R(a, r), kw in R(b, s) (MAS_WLAYOUT_UP2, include/mas_hip.h)"""
    wp = {}
    for a in (0, 1):
        for b in (0, 1):
            for r in (0, 1):
                for s in (0, 1):
                    wp[a, b, r, s] = sum(wt[:, :, kh, kw] for kh in _R[a, r] for kw in _R[b, s])
    return wp


@pytest.mark.parametrize("h,w", [(5, 7), (8, 8), (1, 3)])
def test_upsample_conv_is_four_phase_convolutions(h, w):
    """forward: y[2i + a][2j + b] = sum over (r, s) of Wp[a][b][r][s] x[i + a - 1 + r][j + b - 1 + s] (zero outside the map);
    data gradient: dx[i][j] = sum over (a, b, r, s) of Wp[a][b][r][s]^T dy[2 (i + 1 - a - r) + a][2 (j + 1 - b - s) + b]
    -- the two index maps conv_up2_kernel walks (patch offset (a, b) forward, (1 - a, 1 - b) with flipped taps backward)"""
    n, cin, cout = 2, 3, 4
    x = _rand(n, cin, h, w, seed=11).requires_grad_(True)
    wt = _rand(cout, cin, 3, 3, seed=12)
    y_ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, padding=1)
    dy = _rand(*y_ref.shape, seed=13)
    y_ref.backward(dy)
    wp = _phase_weights(wt)
    xd = x.detach()
    xp = F.pad(xd, (1, 1, 1, 1))                           # xp[i + 1] = x[i]
    y = torch.zeros_like(y_ref)
    for a in (0, 1):
        for b in (0, 1):
            for r in (0, 1):
                for s in (0, 1):
                    win = xp[:, :, a + r:a + r + h, b + s:b + s + w]                    # x[i + a - 1 + r][j + b - 1 + s]
                    y[:, :, a::2, b::2] += torch.einsum("oc,nchw->nohw", wp[a, b, r, s], win)
    assert torch.allclose(y, y_ref.detach(), rtol=1e-12, atol=1e-12)
    dx = torch.zeros_like(xd)
    for a in (0, 1):
        for b in (0, 1):
            dph = F.pad(dy[:, :, a::2, b::2], (1, 1, 1, 1))                             # phase image of dy, dph[i + 1] = D_ab[i]
            for r in (0, 1):
                for s in (0, 1):
                    win = dph[:, :, 2 - a - r:2 - a - r + h, 2 - b - s:2 - b - s + w]   # D_ab[i + 1 - a - r][j + 1 - b - s]
                    dx += torch.einsum("oc,nohw->nchw", wp[a, b, r, s], win)
    assert torch.allclose(dx, x.grad, rtol=1e-12, atol=1e-12)
