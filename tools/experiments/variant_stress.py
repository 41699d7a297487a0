"""round 6: run-to-run stress of VQ-IMG forward + backward in other geometries / modes than the benched one (same harness as misc_stress.py):
MODE=b8 (batch 8, 256^2), r512 (batch 4, 512^2), fp32 (batch 4, 256^2, the exact-fp32 parity mode: conv_fwd.hip's general kernel everywhere),
fused (batch 16, MAS_GN_MATERIALIZE=0 must be exported: GroupNorm+SiLU in the convolutions' loaders)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops                                                                 # noqa: E402
from models import VQBASE                                                               # noqa: E402

mode, reps = os.environ.get("MODE", "b8"), int(os.environ.get("REPS", "300"))
batch, res, dt = {"b8": (8, 256, torch.bfloat16), "r512": (4, 512, torch.bfloat16), "fp32": (4, 256, torch.float32),
                  "fused": (16, 256, torch.bfloat16)}[mode]
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.set_compute_dtype(dt)
m = VQBASE(**bench.IMG_CFG).to(dev).train()
m.quantize.q_counter = m.quantize.q_re_end
x = torch.rand(batch, 3, res, res, generator=torch.Generator().manual_seed(1)).to(dev).requires_grad_(True)
ps = [p for p in m.parameters()]


def isum(t):
    v = t.detach().contiguous()
    return (v.view(torch.int32) if v.dtype == torch.float32 else v.view(torch.int16).to(torch.int32)).sum()


rows = []
for _ in range(reps):
    m.zero_grad(set_to_none=True)
    x.grad = None
    rec, q = m(x)
    ((x - rec).abs().mean() + q).backward()
    rows.append(torch.stack([isum(rec), isum(x.grad)] + [isum(p.grad) for p in ps if p.grad is not None]))
t = torch.stack(rows).cpu()
bad = (t != t[0]).any(dim=1)
cols = (t != t[0]).any(dim=0).nonzero().flatten().tolist()
print(f"VQ-IMG forward + backward, mode {mode} (batch {batch}, {res}^2, {dt}): {reps} repetitions, {int(bad.sum())} differ from the first" + (f"; tensors {cols[:12]} ({len(cols)})" if cols else ""))
