"""mas_hip.optim.Adam on CPU tensors: the host logic (state layout, step counting, hyper-parameter checks, state_dict interchange with
torch.optim.Adam) and the plain-torch update that non-CUDA / non-fp32 parameters take.  The one-launch kernel is tests/test_gpu_adam.py."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in ((5,), (3, 4), (2, 3, 3, 3))]


@pytest.mark.parametrize("wd", [0.0, 0.02])
def test_cpu_parameters_take_the_torch_expression_and_match_torch_adam(wd):
    from mas_hip.optim import Adam
    a, b = _params(1), _params(1)
    oa = Adam(a, lr=2e-3, betas=(0.5, 0.9), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(b, lr=2e-3, betas=(0.5, 0.9), eps=1e-8, weight_decay=wd)
    g = torch.Generator().manual_seed(2)
    for step in range(5):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if step == 2:
            a[1].grad = None
            b[1].grad = None
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
        assert int(oa.state[pa]["step"]) == int(ob.state[pb]["step"])
        assert set(oa.state[pa]) == {"step", "exp_avg", "exp_avg_sq"}
    # state_dict interchange (deep copies: load_state_dict keeps same-device tensors by reference)
    c = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oc = torch.optim.Adam(c, lr=2e-3, betas=(0.5, 0.9), eps=1e-8, weight_decay=wd)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))
    for pa, pc in zip(a, c):
        gr = torch.randn(pa.shape, generator=g)
        pa.grad, pc.grad = gr.clone(), gr.clone()
    oa.step()
    oc.step()
    for pa, pc in zip(a, c):
        assert torch.allclose(pa, pc, rtol=1e-6, atol=1e-7)


def test_constructor_checks():
    from mas_hip.optim import Adam
    p = _params(3)
    for kw in (dict(amsgrad=True), dict(maximize=True), dict(capturable=True), dict(differentiable=True)):
        with pytest.raises(NotImplementedError):
            Adam(p, **kw)
    for kw in (dict(lr=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.5, -0.1)), dict(eps=-1e-8), dict(weight_decay=-0.1)):
        with pytest.raises(ValueError):
            Adam(p, **kw)
    o = Adam(p, lr=1e-3)
    assert o.step() is None and all(len(o.state[q]) == 0 for q in p)          # no gradients: nothing happens
    assert o.step(lambda: torch.tensor(3.0)).item() == 3.0                     # closure result is returned
