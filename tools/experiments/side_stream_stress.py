"""round 6: which tensor is not bitwise reproducible with the weight gradient on the side stream?  (one bench run in ~15 ends on a
different loss.)  Repeats forward + backward of the benched model on the same input and weights; every gradient is compared with the first
repetition's; prints the names that ever differ and how often."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops                                                                 # noqa: E402
from models import VQBASE                                                               # noqa: E402

reps, batch = int(os.environ.get("REPS", "60")), int(os.environ.get("BATCH", "32"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.set_compute_dtype(torch.bfloat16)
m = VQBASE(**bench.IMG_CFG).to(dev).train()
m.quantize.q_counter = m.quantize.q_re_end
x = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev).requires_grad_(True)
names = ["x"] + [n for n, _ in m.named_parameters()]


def once():
    m.zero_grad(set_to_none=True)
    x.grad = None
    rec, q = m(x)
    loss = (x - rec).abs().mean() + q
    loss.backward()
    return [x.grad] + [p.grad for _, p in m.named_parameters()], float(loss)


ref, l0 = once()
ref = [g.clone() if g is not None else None for g in ref]
bad, badloss = {}, 0
for r in range(reps):
    got, l = once()
    badloss += l != l0
    for n, a, b in zip(names, ref, got):
        if a is not None and not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            bad.setdefault(n, []).append((r, float(d.max()), int((d > 0).sum()), a.numel()))
torch.cuda.synchronize()
print(f"MAS_WGRAD_STREAM={os.environ.get('MAS_WGRAD_STREAM', '1')} batch {batch}: {reps} repetitions, {len(bad)} tensors ever differ, forward loss differed {badloss} times")
for n, v in bad.items():
    print(f"  {n}: {len(v)} times; first (rep, max abs diff, elements differing, of) = {v[0]}")
