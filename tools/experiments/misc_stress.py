"""round 6: run-to-run stress of the paths the VQ-IMG bench step does not take: (a) the VQ-SEG training step (159 classes, 256^2, BCE-free L1
stand-in loss), (b) the discriminator's forward + backward (4x4 / stride-2 convolutions on the general kernel of conv_fwd.hip, BatchNorm +
LeakyReLU), (c) the VQ-IMG evaluation forward at batch 4.  Integer checksums on the device, host waits once per workload."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops                                                                 # noqa: E402
from models import VQBASE                                                               # noqa: E402
from losses.discriminator import Discriminator                                          # noqa: E402

reps = int(os.environ.get("REPS", "400"))
dev = torch.device("cuda:0")


def isum(t):
    v = t.detach().contiguous()
    return (v.view(torch.int32) if v.dtype == torch.float32 else v.view(torch.int16).to(torch.int32)).sum()


def stress(label, once, reps):
    rows = []
    for _ in range(reps):
        rows.append(torch.stack([isum(t) for t in once()]))
    t = torch.stack(rows).cpu()
    bad = (t != t[0]).any(dim=1)
    cols = (t != t[0]).any(dim=0).nonzero().flatten().tolist()
    print(f"{label}: {reps} repetitions, {int(bad.sum())} differ from the first" + (f"; tensors {cols[:10]}" if cols else ""))


torch.manual_seed(0)
ops.set_compute_dtype(torch.bfloat16)
# (a) VQ-SEG step
seg = VQBASE(**bench.SEG_CFG).to(dev).train()
seg.quantize.q_counter = seg.quantize.q_re_end
xs = torch.rand(8, 159, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
ps = [p for p in seg.parameters()]


def seg_once():
    seg.zero_grad(set_to_none=True)
    rec, q = seg(xs)
    ((xs - rec).abs().mean() + q).backward()
    return [rec] + [p.grad for p in ps if p.grad is not None]


stress("VQ-SEG forward + backward (B = 8, 159 x 256^2)", seg_once, reps)
del seg
# (b) discriminator
d = Discriminator().to(dev).train()
xd = torch.rand(16, 3, 256, 256, generator=torch.Generator().manual_seed(2)).to(dev).requires_grad_(True)
pd = [p for p in d.parameters()]


def disc_once():
    d.zero_grad(set_to_none=True)
    xd.grad = None
    out = d(xd)
    out.float().mean().backward()
    return [out, xd.grad] + [p.grad for p in pd if p.grad is not None]


stress("discriminator forward + backward (B = 16, 256^2)", disc_once, reps)
# (c) VQ-IMG eval forward
img = VQBASE(**bench.IMG_CFG).to(dev).eval()
xi = torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)


def img_once():
    with torch.no_grad():
        rec, q = img(xi)
    return [rec]


stress("VQ-IMG evaluation forward (B = 4, 256^2)", img_once, reps * 2)
