"""Which Python lines launch the fill kernels of one VQ-IMG training step?  (FillFunctor launches / step < 50)
   python tools/probes/fill_origin.py [--batch 8]   ->  aten::fill_/zero_ calls of ONE step grouped by Python stack."""
import argparse
import collections
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "make-a-scene_amd"))
sys.path.insert(0, R)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from mas_hip import ops
    from models import VQBASE
    import bench
    ops.set_compute_dtype(torch.bfloat16)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = VQBASE(**bench.IMG_CFG).to(dev).train()
    model.quantize.q_counter = model.quantize.q_re_end
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, betas=(0.5, 0.9), fused=True)
    x = torch.rand(a.batch, 3, 256, 256, device=dev)

    def step():
        rec, q = model(x)
        loss = (x - rec).abs().mean() + q
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    groups = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::fill_", "aten::zero_"):
            st = [s for s in (ev.stack or []) if "site-packages/torch" not in s and "<built-in" not in s][:4]
            groups[" <- ".join(s.replace(R + "/", "") for s in st) or "(no python frame: autograd engine)"] += 1
    tot = sum(groups.values())
    print(f"{tot} aten::fill_/zero_ calls in one step")
    for k, v in groups.most_common(25):
        print(f"{v:5d}  {k}")


if __name__ == "__main__":
    main()
