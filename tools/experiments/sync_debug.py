"""round 6: which calls of a training step synchronise the host with the device?  torch.cuda.set_sync_debug_mode("warn") around three steps
of the bench loop; prints each distinct warning site (file:line of the innermost frame inside this repo)."""
import os
import sys
import traceback
import warnings

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops, optim                                                          # noqa: E402
from models import VQBASE                                                               # noqa: E402

dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.bfloat16)
m = VQBASE(**bench.IMG_CFG).to(dev).train()
m.quantize.q_counter = m.quantize.q_re_end
opt = optim.Adam(m.parameters(), lr=5e-6, betas=(0.5, 0.9))
x = torch.rand(32, 3, 256, 256).to(dev)


def step():
    rec, q = m(x)
    loss = (x - rec).abs().mean() + q
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
sites = {}
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))


def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message).lower():
        return
    frames = [f for f in traceback.extract_stack() if f.filename.startswith(root) and "sync_debug.py" not in f.filename]
    key = " <- ".join(f"{os.path.relpath(f.filename, root)}:{f.lineno}" for f in frames[-3:][::-1]) or f"{filename}:{lineno}"
    sites[key] = sites.get(key, 0) + 1


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for _ in range(3):
    step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"synchronising calls in 3 steps: {sum(sites.values())}")
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"  {v:3d} x  {k}")
