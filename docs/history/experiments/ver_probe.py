import torch
for kw in (dict(fused=True), dict(foreach=True), dict(foreach=False, fused=False)):
    p = torch.nn.Parameter(torch.randn(64, 64, device="cuda"))
    p.grad = torch.randn_like(p)
    o = torch.optim.Adam([p], lr=0.1, **kw)
    before = p.detach().clone(); v0 = p._version
    o.step(); torch.cuda.synchronize()
    print(kw, "version", v0, "->", p._version, "changed", bool((p.detach() != before).any()))
import torch.optim.optimizer as oo
print("has global post hook:", hasattr(oo, "register_optimizer_step_post_hook"))
