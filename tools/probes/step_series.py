"""Per-step GPU time of the VQ-IMG training step (HIP events between steps, no host synchronisation inside the loop) for either
optimizer: a periodic slow step would hide in bench.py's average.   OPT=mas|torch B=32 N=40 python tools/probes/step_series.py"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "make-a-scene_amd"))
sys.path.insert(0, R)
from bench import IMG_CFG  # noqa: E402
from mas_hip import ops  # noqa: E402
from models import VQBASE  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = VQBASE(**IMG_CFG)
with torch.no_grad():
    model.quantize.embedding.weight.normal_(0.0, 1.0)
model = model.to(dev).train()
model.quantize.q_counter = model.quantize.q_re_end
if os.environ.get("OPT", "mas") == "mas":
    from mas_hip.optim import Adam
    opt = Adam(model.parameters(), lr=5e-6, betas=(0.5, 0.9))
else:
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, betas=(0.5, 0.9), fused=True)
B, N = int(os.environ.get("B", "32")), int(os.environ.get("N", "40"))
x = torch.rand(B, 3, 256, 256).to(dev)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
for _ in range(5):
    rec, q = model(x); ((x - rec).abs().mean() + q).backward(); opt.step(); opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
evs[0].record()
for i in range(N):
    rec, q = model(x)
    ((x - rec).abs().mean() + q).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print(os.environ.get("OPT", "mas"), "mean %.3f min %.3f max %.3f ms:" % (sum(ts) / N, min(ts), max(ts)), " ".join("%.1f" % t for t in ts))
