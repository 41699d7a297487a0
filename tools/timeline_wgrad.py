#!/usr/bin/env python3
"""Per-tile phase timeline of conv_wgrad from a -DMAS_TIMELINE build (work-group 100, first 4 tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(4 * 8 * 8, dtype=torch.int64, device=dev)
os.environ["MAS_DBG_PTR"] = hex(dbg.data_ptr())
from mas_hip import ops
act = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, c, h = 32, 128, 256
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
dy = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
ss = torch.randn(n, c, 2, device=dev) if act else None
for _ in range(3):
    ops.conv_wgrad_raw(x, ss, dy, n, h, h, c, h, h, c, 3, 1, 1, 1, act, False, True)
torch.cuda.synchronize()
d = dbg.cpu().view(4, 8, 8).double()
if os.environ.get("MAS_WGRAD_NO_TR"):
    names = ["tile top", "barrier0 released", "dY staged", "A staged", "barrier1 released", "MFMA issued"]
else:       # transpose-read kernel: two LDS stages, one barrier per tile
    names = ["tile top", "next loads issued", "MFMA issued", "next tile staged", "barrier released"]
for it in range(1, 4):
    t0 = d[it, :, 0].min()
    print(f"tile {it}: " + "  ".join(f"{names[k]}={int(d[it, :, k].mean() - t0)}" for k in range(len(names))) +
          f"   next tile top={int(d[it + 1, :, 0].mean() - t0) if it < 3 else -1}")
    for wv in range(8):
        print("     wave", wv, [int(d[it, wv, k] - t0) for k in range(len(names))])
