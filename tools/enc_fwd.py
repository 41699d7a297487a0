"""Encoder.forward x N under no_grad (for rocprofv3 kernel traces of the north_star's encoder stack)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd")); sys.path.insert(0, ROOT)
from bench import IMG_CFG
from mas_hip import ops
from models import VQBASE
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = VQBASE(**IMG_CFG).to("cuda").train()
x = torch.rand(int(os.environ.get("B", "32")), 3, 256, 256).cuda()
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", "6"))):
        model.encoder(x)
torch.cuda.synchronize()
