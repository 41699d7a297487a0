"""Per-shape account of the convolution launches of one VQ-IMG training step (B=32, bf16): HIP events around every launch through
ops.set_launch_hook -> count, total ms, TFLOP/s per (kind, shape).  The events serialise nothing (same stream), but each adds ~1 us.
   python tools/conv_shape_profile.py [--batch 32]"""
import argparse
import collections
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "make-a-scene_amd"))
sys.path.insert(0, R)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
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
    rec = {"on": False, "ev": []}

    def hook(kind, shape, launch):
        if not rec["on"]:
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        rec["ev"].append((kind, shape, e0, e1, ops.last_kernel()))

    ops.set_launch_hook(hook)

    def step():
        r, q = model(x)
        loss = (x - r).abs().mean() + q
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rec["on"] = True
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    step()
    t1.record()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for kind, sh, e0, e1, kern in rec["ev"]:
        n, h, w, cin, ho, wo, cout, ks, stride, act, res = sh
        key = (kind, cin, cout, h, ho, ks, stride, act, res, kern)
        ent = agg.setdefault(key, [0, 0.0, 0.0])
        ent[0] += 1
        ent[1] += e0.elapsed_time(e1)
        px = n * ho * wo if kind == "conv_fwd" else n * ho * wo
        ent[2] += 2.0 * ks * ks * cin * cout * px
    tot = sum(v[1] for v in agg.values())
    print(f"step {t0.elapsed_time(t1):.2f} ms; {len(rec['ev'])} conv launches, {tot:.2f} ms inside the hooks")
    print(f"{'kind':10s} {'cin':>4s} {'cout':>4s} {'h':>4s} {'ho':>4s} ks s act res {'calls':>5s} {'ms':>8s} {'ms/call':>8s} {'TF/s':>7s}  last kernel of the launch")
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, cin, cout, h, ho, ks, stride, act, res, kern = key
        print(f"{kind:10s} {cin:4d} {cout:4d} {h:4d} {ho:4d} {ks:2d} {stride:1d} {act:3d} {res:3d} {v[0]:5d} {v[1]:8.3f} {v[1]/v[0]:8.4f} {v[2]/v[1]/1e9:7.0f}  {kern}")


if __name__ == "__main__":
    main()
