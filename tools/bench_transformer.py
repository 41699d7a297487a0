#!/usr/bin/env python3
"""BASELINE config 4: MakeAScene 24L / 1024d / 16 heads over 256 text + 256 seg + 1024 image tokens (S=1536), one GPU.
fwd + bwd of the cross-entropy on the image tokens (reference train.py:150-152), bf16 autocast for the library GEMMs,
HIP flash attention core.  Prints tokens/s and model TFLOP/s (1185.4 GFLOP/sample fwd full-S^2 count)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
from mas_hip import ops
from models.transformer import MakeAScene

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--layers", type=int, default=24)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MakeAScene(a.layers, 1024, 16, 8192, 256, 49408 + 256, 32, 16, 256).to(dev)
text = torch.randint(1, 49408, (a.batch, 256), device=dev); text[:, 200:] = 0
seg = torch.randint(0, 256, (a.batch, 256), device=dev)
img = torch.randint(0, 8192, (a.batch, 1024), device=dev)
def step():
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(text, seg, img)
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, 8192), img.reshape(-1))
    loss.backward()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
gf = 1185.4 * a.layers / 24 * 3 * a.batch
print(f"MakeAScene {a.layers}L/1024d/16h S=1536 B={a.batch} bf16 fwd+bwd: {dt*1e3:.1f} ms/step  {a.batch*1536/dt:.0f} tokens/s  {gf/dt/1e3:.1f} TFLOP/s  loss {float(loss):.3f}")
