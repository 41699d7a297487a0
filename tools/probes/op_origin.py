"""Which Python lines issue a given ATen op (copies, fills, casts) in the no_grad Encoder forward or in one training step?
    python tools/probes/op_origin.py --mode enc_fwd|step [--ops aten::copy_,aten::fill_] [--batch 8]
Groups the calls of ONE pass by Python stack (the rocprof traces show `__amd_rocclr_copyBuffer` / FillFunctor launches without callers)."""
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
    ap.add_argument("--mode", default="enc_fwd", choices=["enc_fwd", "step"])
    ap.add_argument("--ops", default="aten::copy_,aten::fill_,aten::zero_,aten::cat")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    want = set(a.ops.split(","))
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

    def enc():
        with torch.no_grad():
            model.encoder(x)

    fn = step if a.mode == "step" else enc
    for _ in range(3):
        step()
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        fn()
        torch.cuda.synchronize()
    groups = collections.Counter()
    for ev in prof.events():
        if ev.name in want:
            st = [s for s in (ev.stack or []) if "site-packages/torch" not in s and "dist-packages/torch" not in s and "<built-in" not in s][:3]
            where = " <- ".join(s.replace(R + "/", "") for s in st)
            if not where:                      # backward runs on autograd threads (no Python frame): name the enclosing autograd node instead
                par, chain = ev.cpu_parent, []
                while par is not None and len(chain) < 3:
                    chain.append(par.name)
                    par = par.cpu_parent
                where = "(autograd thread) in " + " <- ".join(chain) if chain else "(no python frame, no parent)"
            groups[f"{ev.name} {list(ev.input_shapes)[:2] if ev.input_shapes else ''}  {where}"] += 1
    print(f"{sum(groups.values())} calls of {sorted(want)} in one {a.mode} pass")
    for k, v in groups.most_common(30):
        print(f"{v:5d}  {k}")


if __name__ == "__main__":
    main()
