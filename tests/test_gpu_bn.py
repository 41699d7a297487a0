"""SyncBatchNorm behind quant_conv (reference models/vqvae.py:15-16) on libmas_hip.so's BatchNorm kernels (batchnorm.hip; models.modules.SyncBatchNorm):
against torch.nn.BatchNorm2d in one process (outputs, running statistics over two steps, all gradients, evaluation mode, cumulative
momentum), bitwise run to run, and two ranks sharing the GPU over gloo against the full batch in one process.  (The model-level two-rank
tests of tests/test_gpu_dp.py run through the same exchange.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _pair(c, momentum=0.1, affine=True, seed=0):
    from models.modules import SyncBatchNorm
    torch.manual_seed(seed)
    ours = SyncBatchNorm(c, momentum=momentum, affine=affine).cuda()
    ref = torch.nn.BatchNorm2d(c, momentum=momentum, affine=affine).cuda()
    if affine:
        with torch.no_grad():
            ours.weight.copy_(1.0 + 0.2 * torch.randn(c)); ours.bias.copy_(0.1 * torch.randn(c))
    ref.load_state_dict(ours.state_dict())
    return ours, ref


@pytest.mark.parametrize("shape", [(32, 256, 16, 16), (3, 12, 5, 7), (2, 1024, 4, 4), (1, 32, 1, 2), (5, 64, 33, 9)], ids=lambda s: "x".join(map(str, s)))
def test_training_forward_backward_running_stats_vs_batchnorm2d(shape):
    _dev()
    n, c, h, w = shape
    ours, ref = _pair(c)
    g = torch.Generator().manual_seed(n * 7 + c)
    for step in range(2):
        x = (1.5 * torch.randn(n, c, h, w, generator=g) + 0.7).cuda().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(n, c, h, w, generator=g).cuda()
        xo, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        yo, yr = ours(xo), ref(xr)
        assert yo.is_contiguous(memory_format=torch.channels_last) and yo.dtype == torch.float32
        (yo * gy).sum().backward(); (yr * gy).sum().backward()
        pairs = [("y", yo, yr, 2e-5), ("dx", xo.grad, xr.grad, 1e-4), ("dgamma", ours.weight.grad, ref.weight.grad, 2e-5),
                 ("dbeta", ours.bias.grad, ref.bias.grad, 2e-5)]
        if any(_rel(a, b) >= tol for _, a, b, tol in pairs):
            # ill-conditioned channels (the 1x32x1x2 case: two values per channel -- rstd amplifies the rounding of the variance, xhat is +-1
            # up to that rounding and dx is O(eps / var) of its terms): torch's fp32 statistics and the fp64 sums here round differently;
            # the arbiter is the same module in fp64: ours must be as close to it as torch's fp32 result is (x 1.5), or within the tolerance
            r64 = torch.nn.BatchNorm2d(c, momentum=ref.momentum).double()
            r64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in ours.state_dict().items()})
            x64 = x.detach().cpu().double().requires_grad_(True)
            y64 = r64(x64)
            (y64 * gy.cpu().double()).sum().backward()
            truth = dict(y=y64, dx=x64.grad, dgamma=r64.weight.grad, dbeta=r64.bias.grad)
            for name, a, b, tol in pairs:
                assert _rel(a, truth[name]) <= max(tol, 1.5 * _rel(b, truth[name])), (step, name, _rel(a, truth[name]), _rel(b, truth[name]))
        ours.weight.grad = ours.bias.grad = ref.weight.grad = ref.bias.grad = None
    assert _rel(ours.running_mean, ref.running_mean) < 1e-5 and _rel(ours.running_var, ref.running_var) < 1e-5
    assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 2
    ours.eval(); ref.eval()
    x = torch.randn(n, c, h, w, generator=g).cuda()
    assert _rel(ours(x), ref(x)) < 2e-5
    assert sorted(ours.state_dict().keys()) == sorted(ref.state_dict().keys())


def test_cumulative_momentum_no_affine_and_fallbacks():
    _dev()
    ours, ref = _pair(64, momentum=None, affine=False)
    g = torch.Generator().manual_seed(5)
    for _ in range(3):
        x = torch.randn(4, 64, 8, 8, generator=g).cuda()
        assert _rel(ours(x), ref(x)) < 2e-5
    assert _rel(ours.running_mean, ref.running_mean) < 1e-5 and _rel(ours.running_var, ref.running_var) < 1e-5
    o1, r1 = _pair(32)
    for m in (o1, r1):                                  # one value per channel in training: torch's error, also here
        with pytest.raises(ValueError):
            m(torch.randn(1, 32, 1, 1).cuda())
    o6, r6 = _pair(6)                                   # C % 4 != 0: torch's own implementation takes it
    x = torch.randn(2, 6, 4, 4).cuda()
    assert _rel(o6(x), r6(x)) < 1e-5


def test_eval_mode_is_differentiable_like_batchnorm2d():
    """eval() with frozen statistics must keep x, weight and bias on the autograd graph, as F.batch_norm does."""
    _dev()
    ours, ref = _pair(256)
    g = torch.Generator().manual_seed(3)
    rm, rv = 0.3 * torch.randn(256, generator=g), 0.5 + torch.rand(256, generator=g)      # non-trivial running statistics
    for m in (ours, ref):
        with torch.no_grad():
            m.running_mean.copy_(rm); m.running_var.copy_(rv)
    ours.eval(); ref.eval()
    x = (1.3 * torch.randn(4, 256, 16, 16, generator=g) + 0.2).cuda()
    gy = torch.randn(4, 256, 16, 16, generator=g).cuda()
    xo, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yo, yr = ours(xo), ref(xr)
    assert yo.grad_fn is not None
    (yo * gy).sum().backward(); (yr * gy).sum().backward()
    assert _rel(yo, yr) < 2e-5 and _rel(xo.grad, xr.grad) < 2e-5
    assert _rel(ours.weight.grad, ref.weight.grad) < 2e-5 and _rel(ours.bias.grad, ref.bias.grad) < 2e-5
    # frozen parameters, input gradient only; and no grad at all
    for prm in (ours.weight, ours.bias):
        prm.requires_grad_(False); prm.grad = None
    xo2 = x.clone().requires_grad_(True)
    (ours(xo2) * gy).sum().backward()
    assert torch.equal(xo2.grad, xo.grad) and ours.weight.grad is None
    with torch.no_grad():
        assert ours(x).grad_fn is None


def test_large_channel_mean_keeps_the_variance():
    """|mean| >> std must not cancel in E[x^2] - mean^2 (fp64 sums from the first addition on)."""
    _dev()
    ours, ref = _pair(64)
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(8, 64, 16, 16, generator=g) + 300.0).cuda()
    y = ours(x)
    xd = x.double()
    var = xd.var(dim=(0, 2, 3), unbiased=False)
    mean = xd.mean(dim=(0, 2, 3))
    want = (xd - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + ours.eps) * ours.weight.double()[None, :, None, None] \
        + ours.bias.double()[None, :, None, None]
    assert _rel(y, want) < 2e-4, _rel(y, want)           # (the fp32 input itself carries 300 * 6e-8 = 2e-5 of a standard deviation)
    rv_want = 0.9 + 0.1 * xd.var(dim=(0, 2, 3), unbiased=True)
    assert _rel(ours.running_var, rv_want) < 1e-5, _rel(ours.running_var, rv_want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("slope", [0.2, 1.0])
def test_batch_norm_leaky_relu_vs_torch(dtype, slope):
    """ops.batch_norm_leaky_relu (the BatchNorm2d + LeakyReLU(0.2) pairs of the PatchGAN discriminator, reference
    losses/discriminator.py:26-33) on bf16 / fp32 storage against torch's modules in fp32: training (outputs, all gradients, running
    statistics over two steps) and evaluation (differentiable)."""
    from mas_hip import ops
    _dev()
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    for shape in ((4, 128, 15, 15), (2, 512, 7, 7), (3, 12, 5, 6)):
        n, c, h, w = shape
        torch.manual_seed(c)
        ours, ref = torch.nn.BatchNorm2d(c).cuda(), torch.nn.BatchNorm2d(c).cuda()
        with torch.no_grad():
            ours.weight.copy_(1.0 + 0.2 * torch.randn(c)); ours.bias.copy_(0.1 * torch.randn(c))
        ref.load_state_dict(ours.state_dict())
        act = torch.nn.LeakyReLU(slope) if slope != 1.0 else torch.nn.Identity()
        g = torch.Generator().manual_seed(n + c)
        for step in range(3):
            if step == 2:
                ours.eval(); ref.eval()
            x = (1.5 * torch.randn(n, c, h, w, generator=g) + 0.4).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
            gy = torch.randn(n, c, h, w, generator=g).to(dtype).cuda()
            xo, xr = x.clone().requires_grad_(True), x.float().clone().requires_grad_(True)
            yo = ops.batch_norm_leaky_relu(xo, ours, slope)
            yr = act(ref(xr))
            assert yo.dtype == dtype and yo.is_contiguous(memory_format=torch.channels_last)
            (yo.float() * gy.float()).sum().backward(); (yr * gy.float()).sum().backward()
            assert _rel(yo, yr) < tol and _rel(xo.grad, xr.grad) < 5 * tol, (shape, step, _rel(yo, yr), _rel(xo.grad, xr.grad))
            assert _rel(ours.weight.grad, ref.weight.grad) < tol and _rel(ours.bias.grad, ref.bias.grad) < tol
            ours.weight.grad = ours.bias.grad = ref.weight.grad = ref.bias.grad = None
        assert _rel(ours.running_mean, ref.running_mean) < 1e-5 and _rel(ours.running_var, ref.running_var) < 1e-5
        assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 2


def test_bitwise_run_to_run():
    _dev()
    ours, _ = _pair(256)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, 256, 16, 16, generator=g).cuda()
    gy = torch.randn(32, 256, 16, 16, generator=g).cuda()
    outs = []
    for _ in range(3):
        ours.running_mean.zero_(); ours.running_var.fill_(1.0)
        xi = x.clone().requires_grad_(True)
        y = ours(xi)
        (y * gy).sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone(), ours.weight.grad.clone(), ours.bias.grad.clone(), ours.running_var.clone()))
        ours.weight.grad = ours.bias.grad = None
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))


def _worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MAS_SYNCBN=mode)      # (read when models.modules is imported)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ours, _ = _pair(64)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(6, 64, 9, 5, generator=g) + 0.3).cuda()
    gy = torch.randn(6, 64, 9, 5, generator=g).cuda()
    lo, hi = (0, 2) if rank == 0 else (2, 6)             # uneven split: the exchange carries the counts
    xi = x[lo:hi].clone().requires_grad_(True)
    y = ours(xi)
    (y * gy[lo:hi]).sum().backward()
    gw, gb = ours.weight.grad.clone(), ours.bias.grad.clone()
    dist.all_reduce(gw); dist.all_reduce(gb)            # (sum of the local sums = the full-batch parameter gradients)
    np.savez(out + str(rank), y=y.detach().cpu().numpy(), dx=xi.grad.cpu().numpy(), gw=gw.cpu().numpy(), gb=gb.cpu().numpy(),
             rm=ours.running_mean.cpu().numpy(), rv=ours.running_var.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["2", "1"])
def test_two_ranks_exchange_equals_the_full_batch(tmp_path, mode):
    """SyncBatchNorm's exchange over gloo, two ranks with 2 and 4 images on the one GPU, against torch.nn.BatchNorm2d on all 6 images in
    this process.  MAS_SYNCBN=2: batchnorm.hip with ONE all_reduce of the fp64 {sum, sum of squares, count} forward and of {sum dy,
    sum dy xhat} backward; MAS_SYNCBN=1 (the default): the exchanging case is handed to torch's own module."""
    _dev()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "bn_rank")
    mp.spawn(_worker, args=(2, port, out, mode), nprocs=2, join=True)
    r0, r1 = np.load(out + "0.npz"), np.load(out + "1.npz")
    sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    _, ref = _pair(64)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(6, 64, 9, 5, generator=g) + 0.3).cuda().requires_grad_(True)
    gy = torch.randn(6, 64, 9, 5, generator=g).cuda()
    y = ref(x)
    (y * gy).sum().backward()
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    assert rel(np.concatenate([r0["y"], r1["y"]]), y.detach().cpu().numpy()) < 2e-5
    assert rel(np.concatenate([r0["dx"], r1["dx"]]), x.grad.cpu().numpy()) < 1e-4
    assert rel(r0["gw"], ref.weight.grad.cpu().numpy()) < 2e-5 and rel(r0["gb"], ref.bias.grad.cpu().numpy()) < 2e-5
    for r in (r0, r1):
        assert rel(r["rm"], ref.running_mean.cpu().numpy()) < 1e-5 and rel(r["rv"], ref.running_var.cpu().numpy()) < 1e-5


def _rccl_worker(out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MAS_SYNCBN="2", MAS_SYNCBN_EXCHANGE_AT_WORLD_1="1")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)
    except Exception as e:                                    # no RCCL in this environment: reported, not failed
        open(out, "w").write("skip: " + repr(e))
        return
    torch.cuda.set_device(0)
    from mas_hip import ops
    calls = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        calls.append((t.dtype, tuple(t.shape), t.device.type))
        return real(t, *a, **k)
    dist.all_reduce = spy
    ours, ref = _pair(256)
    g = torch.Generator().manual_seed(4)
    x = (1.2 * torch.randn(8, 256, 16, 16, generator=g) + 0.3).cuda()
    gy = torch.randn(8, 256, 16, 16, generator=g).cuda()
    xo, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yo, yr = ours(xo), ref(xr)
    (yo * gy).sum().backward(); (yr * gy).sum().backward()
    torch.cuda.synchronize()
    res = dict(y=_rel(yo, yr), dx=_rel(xo.grad, xr.grad), dw=_rel(ours.weight.grad, ref.weight.grad), rv=_rel(ours.running_var, ref.running_var),
               calls=calls)
    dist.destroy_process_group()
    open(out, "w").write(repr(res))


def test_exchange_runs_over_rccl_at_world_size_one(tmp_path):
    """The default since round 6 (MAS_SYNCBN=2) exchanges the fp64 sums with torch.distributed.all_reduce on whatever backend the group
    has; two-rank arithmetic is pinned over gloo above.  Here the SAME code path runs on the NCCL (= RCCL) backend -- one rank, the
    exchange forced (MAS_SYNCBN_EXCHANGE_AT_WORLD_1=1) -- so that the fp64 CUDA all_reduce of 2 C + 1 values, its stream ordering
    against the kernels around it and the backward's second exchange have run on RCCL before a multi-GPU node sees them."""
    _dev()
    out = str(tmp_path / "rccl.txt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(out,))
    p.start(); p.join(300)
    assert p.exitcode == 0, p.exitcode
    txt = open(out).read()
    if txt.startswith("skip"):
        pytest.skip(txt)
    res = eval(txt, {"torch": torch})
    assert res["y"] < 2e-5 and res["dx"] < 1e-4 and res["dw"] < 2e-5 and res["rv"] < 1e-5, res
    assert len(res["calls"]) == 2 and all(c == (torch.float64, (513,), "cuda") for c in res["calls"]), res["calls"]
