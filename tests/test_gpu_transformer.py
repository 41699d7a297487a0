"""Transformer rows of the hot path: the flash-style causal attention kernel vs the CPU oracle
of ``SelfAttention.calculate_attention`` (mask*s - (1-mask)*1e4, PB-relax shift, softmax), and the drop-in
``MakeAScene`` vs the golden logits / gradients produced by the reference itself."""
import contextlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relerr(got, ref):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


@pytest.fixture(autouse=True)
def _restore_dtype():
    from mas_hip import ops
    old = ops.compute_dtype()
    yield
    ops.set_compute_dtype(old)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 24, 16), (1, 2, 200, 64), (2, 16, 1536, 64), (1, 3, 130, 32), (2, 2, 333, 64), (1, 1, 65, 64),
                                   (2, 2, 200, 128), (1, 8, 1536, 128), (1, 1, 65, 128)])    # head_dim 128: bf16 forward fast path + generic backward
def test_causal_attention_vs_oracle(shape, dtype):
    from mas_hip import ops
    from oracle import transformer_oracle as TO
    dev = _dev()
    b, h, s, hd = shape
    d = h * hd
    rs = np.random.RandomState(b * 1000 + s)
    qkv = torch.from_numpy(rs.randn(b, s, 3 * d).astype(np.float32))
    if dtype == torch.bfloat16:
        qkv = qkv.bfloat16().float()
    # oracle: reference formula incl. the multiplicative mask with -1e4 fill and the PB-relax shift (transformer.py:44-71)
    ref_in = qkv.clone().requires_grad_(True)
    q, k, v = (t.view(b, s, h, hd).permute(0, 2, 1, 3) for t in torch.split(ref_in, d, dim=-1))
    mask = torch.tril(torch.ones(s, s))[None, None]
    probs = torch.softmax(TO.causal_attention_scores(q, k, mask, hd), dim=-1)
    ref = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, s, d)
    go = torch.from_numpy(rs.randn(b, s, d).astype(np.float32))
    ref.backward(go)
    x = qkv.clone().to(dev).requires_grad_(True)
    out = ops.causal_attention(x, h, dtype=dtype)          # fp32 leaf, arithmetic forced to `dtype`
    out.backward(go.to(dev))
    assert relerr(out, ref) < TOL[dtype]
    assert relerr(x.grad, ref_in.grad) < 2 * TOL[dtype]


def test_make_a_scene_fp32_vs_reference_golden(golden_dir):
    from mas_hip import ops
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "transformer_tiny.npz"))
    cfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40, text_vocab_size=58,
               image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    ops.set_compute_dtype(torch.float32)
    m = MakeAScene(**cfg)
    m.load_state_dict(TO.synth_transformer_state_dict(cfg, seed=5), strict=True)
    m = m.to(dev)
    text, seg, img = (t.to(dev) for t in TO.synth_tokens(cfg, batch=2, seed=5))
    logits = m(text, seg, img)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    assert logits.shape == (2, 16, 96)
    assert relerr(logits, g["logits"]) < 1e-3
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad:"):
            assert relerr(params[k[5:]].grad, g[k]) < 5e-3, k


def test_make_a_scene_autocast_bf16_vs_reference_golden(golden_dir):
    """The mode bench.py --workload transformer / tools/bench_transformer.py time: fp32 parameters and residual stream,
    ``torch.autocast(bfloat16)`` around the forward (bf16 library GEMMs, bf16 attention kernels fed by the bf16 qkv
    projection, fp32->bf16 / bf16->fp32 LayerNorm variants, bf16 GELU).  Logits AND gradients against the reference's own
    fp32 output: logits within 3e-2 of max|logit|, loss within 1e-2, parameter gradients within 6e-2 of their max."""
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "transformer_tiny.npz"))
    cfg = dict(num_layers=2, hidden_dim=64, num_attn_heads=4, image_vocab_size=96, seg_vocab_size=40, text_vocab_size=58,
               image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=8)
    m = MakeAScene(**cfg)
    m.load_state_dict(TO.synth_transformer_state_dict(cfg, seed=5), strict=True)
    m = m.to(dev)
    text, seg, img = (t.to(dev) for t in TO.synth_tokens(cfg, batch=2, seed=5))
    seen = []
    import mas_hip.ops as O
    orig = O._CausalAttention.forward

    def spy(ctx, qkv, n_heads, cd):
        seen.append((qkv.dtype, cd))
        return orig(ctx, qkv, n_heads, cd)
    O._CausalAttention.forward = staticmethod(spy)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = m(text, seg, img)
    finally:
        O._CausalAttention.forward = staticmethod(orig)
    assert seen and all(a == torch.bfloat16 and b == torch.bfloat16 for a, b in seen)      # the bf16 kernels really ran
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), img.reshape(-1))
    loss.backward()
    e_log = relerr(logits, g["logits"])
    print("autocast bf16 MakeAScene: logits relerr %.3e, loss %.5f vs %.5f" % (e_log, float(loss), float(g["loss"])))
    assert e_log < 3e-2
    assert abs(float(loss) - float(g["loss"])) < 1e-2 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad:"):
            e = relerr(params[k[5:]].grad, g[k])
            print("  grad", k[5:], "relerr %.3e" % e)
            assert params[k[5:]].grad.dtype == torch.float32 and e < 6e-2, k


def test_make_a_scene_fp32_input_runs_fp32_attention():
    """No autocast, fp32 parameters (the reference's train_transformer loop, train.py:150): the attention core follows the
    input dtype, i.e. exact-fp32 kernels -- not a silent bf16 downcast."""
    import mas_hip.ops as O
    dev = _dev()
    seen = []
    orig = O._CausalAttention.forward

    def spy(ctx, qkv, n_heads, cd):
        seen.append(cd)
        return orig(ctx, qkv, n_heads, cd)
    O._CausalAttention.forward = staticmethod(spy)
    try:
        O.set_compute_dtype(torch.bfloat16)                  # the conv stack's global default must not leak into the transformer
        x = torch.randn(1, 40, 3 * 64, device=dev)
        y = O.causal_attention(x, 2)
    finally:
        O._CausalAttention.forward = staticmethod(orig)
    assert seen == [torch.float32] and y.dtype == torch.float32


def _gelu_ref(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [8 * 1024 * 33, 100003, 5])
def test_gelu_tanh_fwd_bwd(n, dtype):
    """reference models/transformer.py:11-14; ragged tails (n not a multiple of the 16-byte vector) included"""
    from mas_hip import ops
    dev = _dev()
    rs = np.random.RandomState(n % 97)
    x = torch.from_numpy((3.0 * rs.randn(n)).astype(np.float32)).to(dtype)
    go = torch.from_numpy(rs.randn(n).astype(np.float32)).to(dtype)
    ref_in = x.float().clone().requires_grad_(True)
    ref = _gelu_ref(ref_in)
    ref.backward(go.float())
    xd = x.to(dev).requires_grad_(True)
    y = ops.gelu_tanh(xd)
    y.backward(go.to(dev))
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and relerr(y, ref) < tol
    assert relerr(xd.grad, ref_in.grad) < tol


@pytest.mark.parametrize("case", [
    # rows, D, in dtype, out dtype, residual
    (37, 64, torch.float32, torch.float32, False),
    (37, 64, torch.float32, torch.float32, True),
    (3 * 1536, 1024, torch.float32, torch.bfloat16, False),     # pre-LN under autocast: fp32 residual stream -> bf16 GEMM input
    (3 * 1536, 1024, torch.bfloat16, torch.float32, True),      # sandwich LN: bf16 GEMM output + fp32 residual stream
    (130, 200, torch.bfloat16, torch.bfloat16, True),
    (5, 2048, torch.bfloat16, torch.bfloat16, False),
    (1, 1024, torch.float32, torch.float32, True),
])
def test_layer_norm_fwd_bwd(case):
    """torch.nn.LayerNorm(D, eps=1e-5) [+ residual] (reference models/transformer.py:159-163,197-210) vs torch fp32 on CPU"""
    from mas_hip import ops
    rows, d, tin, tout, has_res = case
    dev = _dev()
    rs = np.random.RandomState(rows + d)
    x = torch.from_numpy((2.0 * rs.randn(rows, d) + 0.5).astype(np.float32)).to(tin)
    w = torch.from_numpy((1.0 + 0.2 * rs.randn(d)).astype(np.float32))
    b = torch.from_numpy((0.1 * rs.randn(d)).astype(np.float32))
    res = torch.from_numpy(rs.randn(rows, d).astype(np.float32)).to(tout) if has_res else None
    go = torch.from_numpy(rs.randn(rows, d).astype(np.float32)).to(tout)
    rx, rw, rb = x.float().clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rres = res.float().clone().requires_grad_(True) if has_res else None
    ref = torch.nn.functional.layer_norm(rx, (d,), rw, rb, 1e-5)
    if has_res:
        ref = ref + rres
    ref.backward(go.float())
    xd = x.to(dev).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    resd = res.to(dev).requires_grad_(True) if has_res else None
    y = ops.layer_norm(xd, wd, bd, 1e-5, resd, out_dtype=tout)
    y.backward(go.to(dev))
    lo = tin == torch.bfloat16 or tout == torch.bfloat16
    assert y.dtype == tout and relerr(y, ref) < (1e-2 if tout == torch.bfloat16 else 2e-6)
    assert relerr(xd.grad, rx.grad) < (1e-2 if lo else 1e-5)
    assert relerr(wd.grad, rw.grad) < (1e-2 if lo else 1e-5)
    assert relerr(bd.grad, rb.grad) < (1e-2 if lo else 1e-5)
    if has_res:
        assert torch.equal(resd.grad.float().cpu(), go.float())


def test_causal_attention_full_size_properties():
    """BASELINE config 4 shapes (16 heads x 64, S = 1536, bf16): rows of the softmax sum to one (V = 1 gives O = 1),
    causality (changing tokens >= t leaves every output < t bitwise unchanged, forward and dq/dk/dv), finiteness."""
    from mas_hip import ops
    dev = _dev()
    b, h, s, hd = 2, 16, 1536, 64
    d = h * hd
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv = torch.randn(b, s, 3 * d, generator=g).bfloat16().to(dev)
    ones = qkv.clone()
    ones[..., 2 * d:] = 1.0
    o = ops.causal_attention(ones, h)
    assert (o.float() - 1.0).abs().max() < 1e-2
    t = 1000
    a = qkv.clone().requires_grad_(True)
    pert = qkv.clone()
    pert[:, t:] = torch.randn(b, s - t, 3 * d, generator=g).bfloat16().to(dev)
    p = pert.requires_grad_(True)
    oa, op_ = ops.causal_attention(a, h), ops.causal_attention(p, h)
    assert torch.equal(oa[:, :t], op_[:, :t]) and torch.isfinite(oa.float()).all()
    go = torch.zeros_like(oa)
    go[:, :t] = torch.randn(b, t, d, generator=g).bfloat16().to(dev)      # only the first t outputs are differentiated
    oa.backward(go)
    op_.backward(go)
    assert torch.equal(a.grad[:, :t], p.grad[:, :t])                      # their gradients cannot see tokens >= t either
    assert float(a.grad[:, t:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,cols,dtype", [(12288, 1024, torch.bfloat16), (12288, 4096, torch.bfloat16), (100, 64, torch.bfloat16),
                                             (1537, 3072, torch.bfloat16), (7, 8, torch.bfloat16), (300, 12, torch.float32),
                                             (4099, 1024, torch.float32)])
def test_colsum_vs_torch(rows, cols, dtype):
    """mas_colsum (the Linear bias gradient) against an fp64 column sum of the same (already rounded) values: fp32 accumulation in
    a fixed order -> within 1e-6 of sum|x| per column, and bitwise identical run to run."""
    from mas_hip import ops
    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, cols, generator=g) + 0.25).to(dtype).cuda()
    got = ops.colsum(x)
    ref = x.double().sum(0)
    scale = x.double().abs().sum(0)
    assert got.dtype == torch.float32 and got.shape == (cols,)
    assert float(((got.double() - ref).abs() / scale).max()) < 1e-6
    assert torch.equal(got, ops.colsum(x))
    with pytest.raises(RuntimeError):
        ops.colsum(x.t())                                  # not contiguous


def test_linear_bf16_node_vs_autocast_nn_linear():
    """models.transformer.Linear under bf16 autocast (one autograd node: library GEMMs, fp32 weight gradient, HIP bias gradient)
    against torch.nn.Linear under the same autocast: outputs and gradients within bf16 rounding of torch's (torch rounds db to
    bf16 before the fp32 cast; ours is an fp32 sum)."""
    from models.transformer import Linear
    torch.manual_seed(3)
    ours = Linear(256, 384).cuda()
    ref = torch.nn.Linear(256, 384).cuda()
    ref.load_state_dict(ours.state_dict())
    x = torch.randn(3, 200, 256, device="cuda")
    xo = x.clone().requires_grad_(True); xr = x.clone().requires_grad_(True)
    gy = torch.randn(3, 200, 384, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yo = ours(xo); yr = ref(xr)
    assert yo.dtype == torch.bfloat16 and yo.grad_fn.__class__.__name__.startswith("_LinearBf16")
    (yo.float() * gy).sum().backward(); (yr.float() * gy).sum().backward()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel(yo, yr) < 4e-3
    assert rel(xo.grad, xr.grad) < 4e-3
    assert ours.weight.grad.dtype == torch.float32 and rel(ours.weight.grad, ref.weight.grad) < 4e-3
    assert rel(ours.bias.grad, ref.bias.grad) < 4e-3
    # against exact fp64 products of the bf16-rounded operands: the fp32 column sum is tighter than torch's bf16 reduction
    xb, gb = x.bfloat16().double().reshape(-1, 256), gy.bfloat16().double().reshape(-1, 384)
    assert rel(ours.weight.grad, gb.t() @ xb) < 4e-3 and rel(ours.bias.grad, gb.sum(0)) < 1e-6
    # outside autocast the layer is nn.Linear
    y32 = ours(x)
    assert y32.dtype == torch.float32 and not y32.grad_fn.__class__.__name__.startswith("_LinearBf16")


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_layer_norm_fork_adds_the_skip_gradient_in_kernel(in_dtype, out_dtype):
    """ops.layer_norm_fork -> (LN(x), x) as one node (mas_layernorm_bwd_add): values and all gradients equal torch's LayerNorm plus
    a separate skip connection, for every dtype pair the transformer uses; an unused output is handled."""
    from mas_hip import ops
    torch.manual_seed(5)
    d = 256
    x = torch.randn(4, 37, d, device="cuda").to(in_dtype)
    w = (1 + 0.1 * torch.randn(d, device="cuda")).requires_grad_(True)
    b = (0.1 * torch.randn(d, device="cuda")).requires_grad_(True)
    ga = torch.randn(4, 37, d, device="cuda"); gs = torch.randn(4, 37, d, device="cuda")
    xo = x.clone().requires_grad_(True)
    y, skip = ops.layer_norm_fork(xo, w, b, 1e-5, out_dtype)
    assert y.dtype == out_dtype and skip.dtype == in_dtype and torch.equal(skip, xo)
    ((y.float() * ga).sum() + (skip.float() * gs).sum()).backward()
    xr = x.float().clone().requires_grad_(True); wr = w.detach().clone().requires_grad_(True); br = b.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5)
    ga_r = ga.to(out_dtype).float() if out_dtype != torch.float32 else ga          # the kernel sees dy in out_dtype
    gs_r = gs.to(in_dtype).float()
    ((yr * ga_r).sum() + (xr * gs_r).sum()).backward()
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm())
    tol = 1e-5 if in_dtype == torch.float32 and out_dtype == torch.float32 else 1e-2
    assert rel(y, yr) < tol and rel(xo.grad, xr.grad) < tol and rel(w.grad, wr.grad) < tol and rel(b.grad, br.grad) < tol
    # only one of the two outputs used
    xo2 = x.clone().requires_grad_(True)
    y2, s2 = ops.layer_norm_fork(xo2, w, b, 1e-5, out_dtype)
    (s2.float() * gs).sum().backward()
    assert rel(xo2.grad, gs_r) < 1e-6
    xo3 = x.clone().requires_grad_(True)
    y3, _ = ops.layer_norm_fork(xo3, w, b, 1e-5, out_dtype)
    (y3.float() * ga).sum().backward()
    xr3 = x.float().clone().requires_grad_(True)
    (torch.nn.functional.layer_norm(xr3, (d,), w.detach(), b.detach(), 1e-5) * ga_r).sum().backward()
    assert rel(xo3.grad, xr3.grad) < tol


@pytest.mark.parametrize("rows,d,tin,tout", [(3 * 1536, 1024, torch.bfloat16, torch.float32), (130, 200, torch.bfloat16, torch.bfloat16),
                                             (37, 64, torch.float32, torch.float32), (4100, 1024, torch.float32, torch.bfloat16)])
def test_layer_norm_bwd_colsum_of_dx(rows, d, tin, tout):
    """mas_layernorm_bwd_colsum: dx / dgamma / dbeta are those of mas_layernorm_bwd_add bit for bit where the work-group count is the same,
    and the extra output is the fp32 column sum of dx AS STORED (within 1e-6 of sum|dx| per column of an fp64 sum), run-to-run bitwise."""
    from mas_hip import ops
    g = torch.Generator().manual_seed(rows + d)
    x = (2.0 * torch.randn(rows, d, generator=g) + 0.5).to(tin).cuda()
    w = (1.0 + 0.2 * torch.randn(d, generator=g)).cuda()
    b = (0.1 * torch.randn(d, generator=g)).cuda()
    go = torch.randn(rows, d, generator=g).to(tout).cuda()
    y = torch.empty(rows, d, dtype=tout, device="cuda")
    mr = torch.empty(rows, 2, dtype=torch.float32, device="cuda")
    L = ops.lib()
    ops.check(L.mas_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), mr.data_ptr(), ops._DT[tin], ops._DT[tout], rows, d, 1e-5,
                                  ops._stream()), "ln fwd")
    dx0, dg0, db0 = ops._layer_norm_bwd(x, w, mr, go, tout, None, False)
    ops._colsum_hint.clear()
    dx1, dg1, db1 = ops._layer_norm_bwd(x, w, mr, go, tout, None, True)
    (ent,) = ops._colsum_hint.slots.values()          # one entry, keyed by (device, stream)
    dc = ent[5]
    assert torch.equal(dx0, dx1)
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm())
    assert rel(dg1, dg0) < 1e-6 and rel(db1, db0) < 1e-6                    # (4 vs 3 work-groups per CU: another summation order)
    ref = dx1.double().sum(0)
    scale = dx1.double().abs().sum(0) + 1e-30
    assert dc.dtype == torch.float32 and dc.shape == (d,)
    assert float(((dc.double() - ref).abs() / scale).max()) < 1e-6
    ops._layer_norm_bwd(x, w, mr, go, tout, None, True)
    assert torch.equal(dc, next(iter(ops._colsum_hint.slots.values()))[5])
    ops._colsum_hint.clear()


def test_linear_bias_gradient_from_the_layer_norm_backward():
    """Linear -> LayerNorm(residual) as the transformer layer composes them (reference transformer.py:201-203): with
    ``producer_bias_grad`` the Linear's bias gradient comes out of the LayerNorm backward kernel (one hit of the hand-off, no
    mas_colsum launch) and equals the separate column sum to fp32 summation-order accuracy; every other gradient is bitwise the
    same.  A tensor that is not the one the LayerNorm wrote (Dropout in between) misses and falls back."""
    from mas_hip import ops
    from models.transformer import LayerNorm, Linear
    torch.manual_seed(11)
    lin, ln = Linear(256, 512).cuda(), LayerNorm(512, eps=1e-5).cuda()
    x = torch.randn(4, 300, 256, device="cuda")
    skip = torch.randn(4, 300, 512, device="cuda")
    gy = torch.randn(4, 300, 512, device="cuda")

    def run(flag, drop=None):
        for p in list(lin.parameters()) + list(ln.parameters()):
            p.grad = None
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t = lin(xi)
            if drop is not None:
                t = drop(t)
            out = ln(t, residual=skip, producer_bias_grad=flag)
        (out.float() * gy).sum().backward()
        return xi.grad.clone(), [p.grad.clone() for p in list(lin.parameters()) + list(ln.parameters())]

    h0 = ops._colsum_hint.hits
    gx0, g0 = run(False)
    assert ops._colsum_hint.hits == h0
    gx1, g1 = run(True)
    assert ops._colsum_hint.hits == h0 + 1 and not ops._colsum_hint.slots
    assert torch.equal(gx0, gx1) and torch.equal(g0[0], g1[0])              # dx, dW: the same tensors went through the same GEMMs
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm())
    assert rel(g1[1], g0[1]) < 1e-6                                          # the bias gradient: another (fixed) summation order
    assert rel(g1[2], g0[2]) < 1e-6 and rel(g1[3], g0[3]) < 1e-6
    torch.manual_seed(1)
    gx2, g2 = run(True, torch.nn.Dropout(0.1))                               # the Linear receives dropout's gradient, not the LayerNorm's
    assert ops._colsum_hint.hits == h0 + 1 and not ops._colsum_hint.slots
    assert torch.isfinite(g2[1]).all()
    # the hand-off itself: sums are given out for the very tensor they were computed from, once, and for nothing else
    a, b = torch.randn(8, 16, device="cuda").bfloat16(), torch.randn(8, 16, device="cuda").bfloat16()
    sums = torch.zeros(16, device="cuda")
    ops._colsum_hint.put(a, sums)
    assert ops._colsum_hint.take(b) is None and not ops._colsum_hint.slots       # another tensor: miss, slot dropped
    ops._colsum_hint.put(a, sums)
    a.add_(1)                                                                          # written since: the version counter moved
    assert ops._colsum_hint.take(a) is None
    ops._colsum_hint.put(a, sums)
    assert ops._colsum_hint.take(a.view(-1, 16)) is sums and ops._colsum_hint.take(a) is None
    ops._colsum_hint.hits = h0 + 1


def test_linear_bf16_shadow_weights_follow_the_optimizer():
    """The bf16 copies of the Linear parameters are refreshed together when an optimizer step (or any in-place write that bumps
    ``_version``) made them stale, reused between steps, and dropped by invalidate_weight_cache()."""
    from mas_hip import ops
    from models.transformer import Linear
    torch.manual_seed(9)
    a, b = Linear(64, 64).cuda(), Linear(64, 128).cuda()
    x = torch.randn(5, 64, device="cuda")
    opt = torch.optim.SGD(list(a.parameters()) + list(b.parameters()), lr=0.5)
    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return b(a(x)).float()
    y0 = run()
    sa = ops._bf16_shadows.get(a.weight)
    assert torch.equal(sa, a.weight.detach().bfloat16()) and ops._bf16_shadows.get(a.weight) is sa      # reused while the version stands
    y0.square().mean().backward()
    opt.step()
    y1 = run()
    assert not torch.equal(y0, y1)
    for lin in (a, b):
        assert torch.equal(ops._bf16_shadows.get(lin.weight), lin.weight.detach().bfloat16())
        assert torch.equal(ops._bf16_shadows.get(lin.bias), lin.bias.detach().bfloat16())
    ref = torch.nn.functional.linear(torch.nn.functional.linear(x.bfloat16(), a.weight.bfloat16(), a.bias.bfloat16()), b.weight.bfloat16(), b.bias.bfloat16())
    assert float((y1 - ref.float()).abs().max()) < 2e-2 * float(ref.float().abs().max())
    with torch.no_grad():
        a.weight.data.mul_(2.0)                      # a write through .data: invisible to the version check
    ops.invalidate_weight_cache()
    assert torch.equal(ops._bf16_shadows.get(a.weight), a.weight.detach().bfloat16())


@pytest.mark.parametrize("mode", ["autocast", "fp32"])
@pytest.mark.parametrize("rows,d", [(300, 1024), (37, 64), (1030, 256)])
def test_layer_norm_pair_equals_the_two_separate_launches(mode, rows, d):
    """ops.layer_norm_pair (round 6: sandwich LayerNorm + residual and the next pre-LayerNorm as one launch each way; reference
    transformer.py:201-205, 207-209 + the next layer's :197) against the two nodes it replaces (``layer_norm`` with a residual, then
    ``layer_norm_fork``): outputs, dh and dres bit for bit; the four parameter gradients and the Linear's bias gradient to summation
    order; and against torch's fp32 LayerNorm arithmetic."""
    import torch.nn.functional as F
    from mas_hip import ops
    from models.transformer import LayerNorm, Linear
    torch.manual_seed(rows + d)
    lin = Linear(64, d).cuda()
    ln1, ln2 = LayerNorm(d, eps=1e-5).cuda(), LayerNorm(d, eps=1e-5).cuda()
    with torch.no_grad():
        for m in (ln1, ln2):
            m.weight.copy_(1.0 + 0.2 * torch.randn(d)); m.bias.copy_(0.1 * torch.randn(d))
    x = torch.randn(rows, 64, device="cuda")
    res = torch.randn(rows, d, device="cuda")
    gy, gs = torch.randn(rows, d, device="cuda"), torch.randn(rows, d, device="cuda")
    params = list(lin.parameters()) + list(ln1.parameters()) + list(ln2.parameters())

    def run(paired):
        for p in params:
            p.grad = None
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        ctx = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "autocast" else contextlib.nullcontext()
        with ctx:
            h = lin(xi)
            if paired:
                y2, xnew = ops.layer_norm_pair(h, ri, ln1, ln2, producer_bias_grad=True)
            else:
                xn = ln1(h, residual=ri, producer_bias_grad=True)
                y2, xnew = ln2.fork(xn)
        ((y2.float() * gy).sum() + (xnew * gs).sum()).backward()
        return y2.detach(), xnew.detach(), xi.grad.clone(), ri.grad.clone(), [p.grad.clone() for p in params]

    old = ops._LN_PAIR
    try:
        ops._LN_PAIR = True
        h0 = ops._colsum_hint.hits
        y_p, x_p, gx_p, gr_p, gp_p = run(True)
        hits_pair = ops._colsum_hint.hits - h0
        y_s, x_s, gx_s, gr_s, gp_s = run(False)
    finally:
        ops._LN_PAIR = old
    assert y_p.dtype == (torch.bfloat16 if mode == "autocast" else torch.float32) and x_p.dtype == torch.float32
    assert torch.equal(y_p, y_s) and torch.equal(x_p, x_s)
    assert torch.equal(gr_p, gr_s) and torch.equal(gx_p, gx_s)                   # dres; dx through the same GEMM from a bitwise-equal dh
    assert torch.equal(gp_p[0], gp_s[0])                                         # the Linear's weight gradient: same dh
    rel = lambda a, c: float((a.double() - c.double()).norm() / (c.double().norm() + 1e-30))
    for a, c in zip(gp_p[1:], gp_s[1:]):                                         # bias gradient + the four LayerNorm parameter gradients
        assert rel(a, c) < 2e-6, rel(a, c)
    if mode == "autocast":
        assert hits_pair == 1                                                    # the Linear took its bias gradient from the pair's backward
    # and the arithmetic itself, against torch in fp32 on the same h
    with torch.no_grad():
        h = lin(x) if mode == "fp32" else F.linear(x.bfloat16(), lin.weight.bfloat16(), lin.bias.bfloat16()).float()
        xr = res + F.layer_norm(h, (d,), ln1.weight, ln1.bias, 1e-5)
        yr = F.layer_norm(xr, (d,), ln2.weight, ln2.bias, 1e-5)
    tol = 2e-5 if mode == "fp32" else 2e-2
    assert rel(x_p, xr) < tol and rel(y_p.float(), yr) < tol


@pytest.mark.parametrize("rows,cols", [(12288, 4096), (300, 4096), (96, 3072), (64, 640), (5, 64)])
def test_gelu_backward_leaves_the_linear_bias_gradient(rows, cols):
    """Linear -> tanh-GELU (the MLP's lin1, reference transformer.py:125,129): the GELU backward kernel also sums the dx it writes
    (``mas_gelu_tanh_bwd_colsum``) and the Linear takes its bias gradient from there -- one hit of the hand-off, dx and every other gradient
    bit for bit those of the plain kernel + ``mas_colsum``, the bias gradient equal to summation order; and the raw entry point against an
    fp64 column sum of the dx it wrote."""
    from mas_hip import ops
    from models.transformer import Linear
    torch.manual_seed(rows + cols)
    lin = Linear(64, cols).cuda()
    x = torch.randn(rows, 64, device="cuda")
    gy = torch.randn(rows, cols, device="cuda")

    def run(on):
        old, ops._GELU_COLSUM = ops._GELU_COLSUM, on
        try:
            for p in lin.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = ops.gelu_tanh(lin(xi))
            h0 = ops._colsum_hint.hits
            (y.float() * gy).sum().backward()
            return y.detach(), xi.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone(), ops._colsum_hint.hits - h0
        finally:
            ops._GELU_COLSUM = old
    y1, gx1, gw1, gb1, hits1 = run(True)
    y0, gx0, gw0, gb0, hits0 = run(False)
    assert hits1 == 1 and hits0 == 0
    rel = lambda a, c: float((a.double() - c.double()).norm() / (c.double().norm() + 1e-30))
    # (the two instantiations of the kernel may contract their fp32 multiply-adds differently: a rare last-bit flip of a bf16 dx, not more)
    assert torch.equal(y1, y0) and rel(gx1, gx0) < 1e-3 and rel(gw1, gw0) < 1e-3
    assert rel(gb1, gb0) < 1e-3, rel(gb1, gb0)
    # the raw entry point
    L = ops.lib()
    a = torch.randn(rows, cols, device="cuda").bfloat16()
    g = torch.randn(rows, cols, device="cuda").bfloat16()
    dx, dx_ref = torch.empty_like(a), torch.empty_like(a)
    dc = torch.empty(cols, dtype=torch.float32, device="cuda")
    wsb = L.mas_gelu_tanh_bwd_colsum_workspace(ops._DT[a.dtype], cols)
    assert wsb > 0
    ws = torch.empty(wsb // 4, dtype=torch.float32, device="cuda")
    ops.check(L.mas_gelu_tanh_bwd_colsum(a.data_ptr(), g.data_ptr(), dx.data_ptr(), dc.data_ptr(), ops._DT[a.dtype], rows, cols, ws.data_ptr(), wsb,
                                         ops._stream()), "gelu bwd colsum")
    ops.check(L.mas_gelu_tanh_bwd(a.data_ptr(), g.data_ptr(), dx_ref.data_ptr(), ops._DT[a.dtype], a.numel(), ops._stream()), "gelu bwd")
    diff = (dx.float() - dx_ref.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(dx_ref.float().abs().max()) and float((diff > 0).float().mean()) < 1e-2
    ref = dx.double().sum(0)
    scale = dx.double().abs().sum(0) + 1e-30
    assert float(((dc.double() - ref).abs() / scale).max()) < 1e-6


@pytest.mark.parametrize("hidden,heads", [(96, 4), (192, 2), (80, 5), (100, 5), (320, 2)], ids=["hd24", "hd96", "hd16x5", "hd20", "hd160"])
@pytest.mark.parametrize("mode", ["fp32", "autocast"])
def test_make_a_scene_off_config_widths_vs_oracle(hidden, heads, mode):
    """Head dimensions / hidden sizes nobody tuned a kernel for (the reference's constructor accepts any hidden_dim divisible by the head
    count, models/transformer.py:17-35): forward + backward of a small MakeAScene against the oracle.  Whatever the attention / LayerNorm /
    GELU dispatch does with such widths (HIP kernel, padded kernel, ATen on the GPU), the numbers must be the reference's."""
    from mas_hip import ops
    from models.transformer import MakeAScene
    from oracle import transformer_oracle as TO
    dev = _dev()
    cfg = dict(num_layers=2, hidden_dim=hidden, num_attn_heads=heads, image_vocab_size=70, seg_vocab_size=30, text_vocab_size=45,
               image_tokens_per_dim=4, seg_tokens_per_dim=2, text_length=7)
    sd = TO.synth_transformer_state_dict(cfg, seed=9)
    text, seg, img = TO.synth_tokens(cfg, batch=3, seed=9)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref = TO.make_a_scene_forward(sdr, cfg, text, seg, img)
    ref_loss = torch.nn.functional.cross_entropy(ref.reshape(-1, ref.shape[-1]), img.reshape(-1))
    ref_loss.backward()
    ops.set_compute_dtype(torch.float32)
    m = MakeAScene(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "autocast" else contextlib.nullcontext()
    with ctx:
        logits = m(text.to(dev), seg.to(dev), img.to(dev))
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), img.to(dev).reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    tol = 1e-3 if mode == "fp32" else 4e-2
    assert relerr(logits.float(), ref.detach()) < tol
    assert abs(float(loss) - float(ref_loss)) < (1e-4 if mode == "fp32" else 2e-2) * abs(float(ref_loss))
    params = dict(m.named_parameters())
    for k in ("transformer.layers.0.attn.qkv.weight", "transformer.layers.1.mlp.lin1.bias", "transformer.layers.1.first_ln_sandwich.weight",
              "to_logits.1.weight"):
        assert relerr(params[k].grad, sdr[k].grad) < (5e-3 if mode == "fp32" else 8e-2), k
