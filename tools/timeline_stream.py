#!/usr/bin/env python3
"""Per-wave stage timeline of the stream-scheduled 3x3 kernel (a -DS_TIMELINE build: tools/build_file_variant.sh s_tl conv3x3_stream.hip
-DS_TIMELINE; run with MAS_HIP_LIB=<that .so>).  Work-group 100, its 2nd and 3rd tiles.  Stamps per stage s: arrive at the
wait+barrier, barrier released, DMA / deferred store issued; the MFMA block is (next arrive) - (issued)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(2 * 8 * 64, dtype=torch.int64, device=dev)
os.environ["MAS_DBG_PTR"] = hex(dbg.data_ptr())
from mas_hip import ops
act = int(sys.argv[1]) if len(sys.argv) > 1 else 0
res = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, c, h = 32, 128, 256
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
wp = ops.pack_conv_weight(w, False, torch.bfloat16) if os.environ.get('TL_KERNEL', 'stream') == 'stream' else ops.ConvWeight(torch.nn.Parameter(w), False)
ss = torch.randn(n, c, 2, device=dev) if act else None
r = torch.randn_like(x) if res else None
b = torch.randn(c, device=dev)
for _ in range(3):
    ops.conv_fwd_raw(x, ss, wp, b, r, n, h, h, c, h, h, c, 3, 1, 1, 1, act, False, torch.bfloat16)
torch.cuda.synchronize()
d = dbg.cpu().view(2, 8, 64)
for it in range(2):
    t0 = int(d[it, :, 0].min())
    rel = lambda i: [int(v) - t0 for v in d[it, :, i]]
    print(f"--- tile {it + 1} of work-group 100 (act={act} res={res}); cycles since the earliest wave's tile start; waves 0..7")
    print("tile start      ", rel(0))
    prev_issue = None
    tot = {"wait": 0, "issue": 0, "mfma": 0}
    wide = os.environ.get('TL_KERNEL', 'stream') != 'stream'
    nst = 12 if wide else 9
    for s in range(nst):
        a, bb, cc = rel(1 + 3 * s), rel(2 + 3 * s), rel(3 + 3 * s)
        mean = lambda v: sum(v) / len(v)
        line = f"stage {s}: arrive {int(mean(a)):6d} (spread {max(a) - min(a):5d})  barrier+wait {int(mean(bb) - mean(a)):5d}  issue {int(mean(cc) - mean(bb)):5d}"
        if prev_issue is not None:
            line += f"  | mfma block of stage {s - 1}: {int(mean(a) - prev_issue):5d}"
            tot["mfma"] += mean(a) - prev_issue
        tot["wait"] += mean(bb) - mean(a); tot["issue"] += mean(cc) - mean(bb)
        prev_issue = mean(cc)
        print(line)
    e0, e1 = (rel(60), rel(61)) if wide else (rel(40), rel(41))
    print(f"mfma block of the last stage: {int(sum(e0) / 8 - prev_issue)}   epilogue {int(sum(e1) / 8 - sum(e0) / 8)}  tile total {int(sum(e1) / 8 - sum(rel(0)) / 8)}")
    print({k: int(v) for k, v in tot.items()})
