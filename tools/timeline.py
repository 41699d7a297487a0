#!/usr/bin/env python3
"""Prints the per-wave phase timeline recorded by a -DMAS_TIMELINE build of conv_fwd (work-group 100, first 2 tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(2 * 8 * 64, dtype=torch.int64, device=dev)
os.environ["MAS_DBG_PTR"] = hex(dbg.data_ptr())
from mas_hip import ops
act = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, c, h = 32, 128, 256
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
wp = ops.pack_conv_weight(w, False, torch.bfloat16)
ss = torch.randn(n, c, 2, device=dev) if act else None
for _ in range(3):
    ops.conv_fwd_raw(x, ss, wp, None, None, n, h, h, c, h, h, c, 3, 1, 1, 1, act, False, torch.bfloat16)
torch.cuda.synchronize()
nw = 4 if os.environ.get("MAS_CONV_SMALL_TILE") else 8
d = dbg.cpu()[: 2 * nw * 64].view(2, nw, 64)
for it in range(2):
    t0 = int(d[it, :nw, 0].min())
    print(f"--- tile iteration {it}: per tap [mfma-issue-done .. w-commit-done .. barrier-released], cycles since tile start, waves 0..{nw-1}")
    print("tile start / first sync", [int(x) - t0 for x in d[it, :nw, 0]], [int(x) - t0 for x in d[it, :nw, 1]])
    prevC = None
    for ch in range(2):
        for tap in range(9 if nw == 4 else 3):
            a, b, c = (d[it, :nw, 2 + ch * 28 + tap * 3 + k] for k in range(3))
            if (c == 0).all():
                continue
            A, B, C = int(a.double().mean()) - t0, int(b.double().mean()) - t0, int(c.double().mean()) - t0
            extra = f"  tap body(prev barrier->mfma issued)={A - prevC:5d}" if prevC is not None else ""
            print(f"c{ch} tap{tap}: A={A:6d} wait+commit={B - A:5d} barrier={C - B:5d}{extra}")
            prevC = C
    print("epilogue start/end", int(d[it, :nw, 60].double().mean()) - t0, int(d[it, :nw, 61].double().mean()) - t0)
