"""mas_hip.optim.Adam (optim.hip: every parameter in one launch) against torch.optim.Adam -- the reference's optimizer, train.py:99-103:
same update over several steps (fused and single-tensor torch implementations), weight decay, tensor sizes around the 4096-element
block and the 16-byte vector path, state_dict interchange in both directions, parameters without a gradient."""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))

SIZES = [(1,), (7,), (128,), (4096,), (4097,), (3, 5, 7), (100003,), (512, 512, 3, 3), (8191,), (2, 4096)]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in SIZES]


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("wd", [0.0, 0.01])
@pytest.mark.parametrize("torch_kw", [dict(fused=True), dict(foreach=False, fused=False)])
def test_matches_torch_adam_over_several_steps(wd, torch_kw):
    from mas_hip.optim import Adam
    dev = _dev()
    ours, theirs = _params(dev, 3), _params(dev, 3)
    # an unaligned view as one more parameter: storage offset 4 bytes -> the element-wise path of the kernel
    base_o, base_t = torch.randn(1001, device=dev), None
    base_t = base_o.clone()
    ours.append(torch.nn.Parameter(base_o[1:]))
    theirs.append(torch.nn.Parameter(base_t[1:]))
    o = Adam(ours, lr=3e-3, betas=(0.5, 0.9), eps=1e-8, weight_decay=wd)
    t = torch.optim.Adam(theirs, lr=3e-3, betas=(0.5, 0.9), eps=1e-8, weight_decay=wd, **torch_kw)
    g = torch.Generator().manual_seed(5)
    for step in range(6):
        for po, pt in zip(ours, theirs):
            gr = torch.randn(po.shape, generator=g).to(dev) * (0.1 + step)
            po.grad, pt.grad = gr.clone(), gr.clone()
        if step == 3:                       # a parameter that sits a step out (no gradient)
            ours[2].grad = None
            theirs[2].grad = None
        o.step()
        t.step()
    for i, (po, pt) in enumerate(zip(ours, theirs)):
        assert _rel(po.detach(), pt.detach()) < 2e-6, (i, tuple(po.shape), _rel(po.detach(), pt.detach()))
        so, st = o.state[po], t.state[pt]
        assert _rel(so["exp_avg"], st["exp_avg"]) < 2e-6 and _rel(so["exp_avg_sq"], st["exp_avg_sq"]) < 2e-6
        assert int(so["step"]) == int(st["step"])
    from mas_hip import ops
    assert ops.last_kernel() == "adam_multi"


def test_state_dict_interchange_with_torch_adam():
    from mas_hip.optim import Adam
    dev = _dev()
    a, b = _params(dev, 9), _params(dev, 9)
    oa = Adam(a, lr=1e-3, betas=(0.5, 0.9))
    ob = torch.optim.Adam(b, lr=1e-3, betas=(0.5, 0.9), fused=True)
    g = torch.Generator().manual_seed(1)

    def grads():
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(dev)
            pa.grad, pb.grad = gr.clone(), gr.clone()
    for _ in range(2):
        grads(); oa.step(); ob.step()
    # ours -> torch's, torch's -> ours, then one more step everywhere
    c, d = [torch.nn.Parameter(p.detach().clone()) for p in a], [torch.nn.Parameter(p.detach().clone()) for p in b]
    oc = torch.optim.Adam(c, lr=1e-3, betas=(0.5, 0.9), fused=True)
    od = Adam(d, lr=1e-3, betas=(0.5, 0.9))
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))     # (deep copies: load_state_dict keeps same-device tensors by reference, and the
    od.load_state_dict(copy.deepcopy(ob.state_dict()))     #  source optimizers take another step below)
    grads()
    for pc, pd, pa in zip(c, d, a):
        pc.grad, pd.grad = pa.grad.clone(), pa.grad.clone()
    oa.step(); ob.step(); oc.step(); od.step()
    for pa, pb, pc, pd in zip(a, b, c, d):
        assert _rel(pc.detach(), pa.detach()) < 2e-6 and _rel(pd.detach(), pb.detach()) < 2e-6 and _rel(pa.detach(), pb.detach()) < 2e-6


def test_unsupported_variants_raise():
    from mas_hip.optim import Adam
    dev = _dev()
    p = [torch.nn.Parameter(torch.zeros(4, device=dev))]
    for kw in (dict(amsgrad=True), dict(maximize=True), dict(capturable=True)):
        with pytest.raises(NotImplementedError):
            Adam(p, **kw)


def test_edge_cases_empty_parameter_device_step_and_mixed_step_counts():
    """(1) a parameter of numel 0 with a gradient is accepted, as torch.optim.Adam accepts it; (2) a state whose `step`
    lives on the device (loaded from torch.optim.Adam(fused=True)) is moved to the host once; (3) parameters of one group on
    different step counts each take the kernel with their own bias corrections (no fall-back to the slow path for all but the first)."""
    from mas_hip.optim import Adam
    dev = _dev()
    torch.manual_seed(0)
    empty = torch.nn.Parameter(torch.zeros(0, device=dev))
    empty.grad = torch.zeros(0, device=dev)
    opt = Adam([empty], lr=1e-2)
    opt.step()                                               # must not raise ("empty batch")
    # mixed step counts: b joins after a has taken two steps
    a0, b0 = torch.randn(1000, device=dev), torch.randn(777, device=dev)
    ga, gb = torch.randn(1000, device=dev), torch.randn(777, device=dev)
    a, b = torch.nn.Parameter(a0.clone()), torch.nn.Parameter(b0.clone())
    ra, rb = torch.nn.Parameter(a0.clone()), torch.nn.Parameter(b0.clone())
    opt, ref = Adam([a, b], lr=1e-2), torch.optim.Adam([ra, rb], lr=1e-2)
    for k in range(5):
        a.grad, ra.grad = ga * (k + 1), ga * (k + 1)
        if k >= 2:
            b.grad, rb.grad = gb * (k + 1), gb * (k + 1)
        opt.step()
        ref.step()
    torch.cuda.synchronize()
    assert torch.allclose(a, ra, rtol=2e-6, atol=2e-6) and torch.allclose(b, rb, rtol=2e-6, atol=2e-6)
    assert int(opt.state[a]["step"]) == 5 and int(opt.state[b]["step"]) == 3
    # device-resident `step` (fused torch Adam's state layout)
    fa = torch.nn.Parameter(a0.clone())
    fused = torch.optim.Adam([fa], lr=1e-2, fused=True)
    fa.grad = ga.clone()
    fused.step()
    ma = torch.nn.Parameter(fa.detach().clone())
    mine = Adam([ma], lr=1e-2)
    mine.load_state_dict(copy.deepcopy(fused.state_dict()))          # (load_state_dict keeps tensors of matching dtype / device: no aliasing)
    ma.grad, fa.grad = ga * 2, ga * 2
    mine.step()
    fused.step()
    torch.cuda.synchronize()
    assert mine.state[ma]["step"].device.type == "cpu" and int(mine.state[ma]["step"]) == 2
    assert torch.allclose(ma, fa, rtol=2e-6, atol=2e-6)
