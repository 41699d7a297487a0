"""How much of a training step is host enqueue time?  Prints host-side enqueue ms/step (no syncs
inside the loop) next to the wall ms/step, for the whole step and for Encoder.forward alone."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd")); sys.path.insert(0, ROOT)
from bench import IMG_CFG
from mas_hip import ops
from models import VQBASE

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
B = int(os.environ.get("B", "32"))
x = torch.rand(B, 3, 256, 256).to(dev)

def step():
    rec, q = model(x)
    loss = (x - rec).abs().mean() + q
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)

for _ in range(3):
    step()
torch.cuda.synchronize()
K = 6
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} train step: host enqueue {1e3*(t1-t0)/K:.2f} ms/step, wall {1e3*(t2-t0)/K:.2f} ms/step")
with torch.no_grad():
    for _ in range(2):
        model.encoder(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        model.encoder(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"B={B} encoder fwd: host enqueue {1e3*(t1-t0)/K:.2f} ms, wall {1e3*(t2-t0)/K:.2f} ms")
