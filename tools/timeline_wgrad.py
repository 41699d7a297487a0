#!/usr/bin/env python3
"""Per-wave row timeline of the LDS-DMA wgrad kernel (a -DD_TIMELINE build: tools/build_file_variant.sh wg_tl conv_wgrad_dma.hip
-DD_TIMELINE; run with MAS_HIP_LIB=<that .so>).  Work-group 100, its tiles 40 and 41; s_memtime ticks (shader cycles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(2 * 8 * 16, dtype=torch.int64, device=dev)
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
d = dbg.cpu().view(2, 8, 16)
for it in range(2):
    t0 = int(d[it, :, 0].min())
    print(f"--- tile {40 + it} of work-group 100 (act={act}); cycles since the earliest wave's arrival; waves 0..7")
    names = ["arrive", "released"] + [f"row {r} issued" for r in range(10)]
    prev = None
    for k, nm in enumerate(names):
        v = [int(t) - t0 for t in d[it, :, k]]
        m = sum(v) / 8
        print(f"{nm:14s} mean {int(m):6d}  d {int(m - prev) if prev is not None else 0:5d}   {v}")
        prev = m
    if it == 0:
        print("tile period (arrive -> next arrive), per wave:", [int(d[1, w, 0] - d[0, w, 0]) for w in range(8)])
