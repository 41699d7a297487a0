"""round 6: forward-only repetition of the benched model with an integer checksum of every module output (taken on the device; the host
waits once at the end): which module's output is not bitwise reproducible, and how often?"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops                                                                 # noqa: E402
from models import VQBASE                                                               # noqa: E402

reps, batch = int(os.environ.get("REPS", "1500")), int(os.environ.get("BATCH", "32"))
grad = os.environ.get("GRAD", "1") == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.set_compute_dtype(torch.bfloat16)
m = VQBASE(**bench.IMG_CFG).to(dev).train()
m.quantize.q_counter = m.quantize.q_re_end
x = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
names, sums = [], []


def hook(mod, inp, out, name=None):
    o = out[0] if isinstance(out, (tuple, list)) else out
    if torch.is_tensor(o) and o.is_cuda and o.dtype in (torch.float32, torch.bfloat16):
        v = o.detach().contiguous()
        v = v.view(torch.int32) if v.dtype == torch.float32 else v.view(torch.int16).to(torch.int32)
        sums.append(v.sum())
        if len(names) < 400:
            names.append(name)


for n_, mod in m.named_modules():
    if n_:
        mod.register_forward_hook(lambda a, b, c, n_=n_: hook(a, b, c, n_))
rows = []
with torch.set_grad_enabled(grad):
    for r in range(reps):
        sums.clear()
        m(x)
        rows.append(torch.stack(sums))
        if r == 0:
            per = len(sums)
t = torch.stack(rows).cpu()
bad = (t != t[0]).nonzero().tolist()
first = {}
for r, c in bad:
    first.setdefault(r, c)
hist = {}
for r, c in first.items():
    hist[names[c]] = hist.get(names[c], 0) + 1
print(f"forward only, grad={int(grad)}, batch {batch}: {reps} repetitions, {len(first)} differ from the first; first differing module per repetition: {hist}")
