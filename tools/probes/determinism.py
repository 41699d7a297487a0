import sys, torch
sys.path.insert(0, "make-a-scene_amd"); sys.path.insert(0, ".")
import bench
from mas_hip import ops
from models import VQBASE
ops.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQBASE(**bench.IMG_CFG)
with torch.no_grad():
    m.quantize.embedding.weight.normal_(0.0, 1.0)
m = m.to(dev).train(); m.quantize.q_counter = m.quantize.q_re_end
x = torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
def grads():
    m.zero_grad(set_to_none=True)
    rec, q = m(x); ((x - rec).abs().mean() + q).backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
a = grads(); b = grads(); c = grads()
bad = [k for k in a if not (torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]))]
print("parameters with gradients:", len(a), "| not bitwise run to run:", len(bad))
for k in bad[:20]:
    print("  ", k, float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30)))
