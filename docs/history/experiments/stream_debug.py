#!/usr/bin/env python3
"""Debug aid for conv3x3_stream.hip: identity / single-tap weights make the output a shifted copy of the input, so a wrong
placement shows up as WHERE each output value came from.  MAS_CONV_STREAM_MIN_TILES_PER_CU=0 python tools/stream_debug.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import numpy as np, torch
from mas_hip import ops
dev = torch.device("cuda:0")
n, c, h, w = 1, 128, 16, 16
# x[p][ch] = unique value encodable in bf16? use small integers: pixel index in one tensor, channel in another
def run(x, wt):
    wp = ops.pack_conv_weight(wt.to(dev), False, torch.bfloat16)
    xd = x.bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd_raw(xd, None, wp, None, None, n, h, w, c, h, w, c, 3, 1, 1, 1, 0, False, torch.bfloat16)
    torch.cuda.synchronize()
    return y.float().cpu()
ii, jj = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
xpix = (ii * 16 + jj).float()[None, None].expand(1, c, h, w).contiguous()          # value = pixel index (0..255, exact in bf16)
xch = torch.arange(c).float()[None, :, None, None].expand(1, c, h, w).contiguous()  # value = channel index
for tap in (4, 0, 8):
    kh, kw = tap // 3, tap % 3
    wt = torch.zeros(c, c, 3, 3); wt[torch.arange(c), torch.arange(c), kh, kw] = 1.0
    yp, yc = run(xpix, wt), run(xch, wt)
    print(f"== identity weights at tap ({kh},{kw}): y[pixel][cout] should be x[pixel+({kh-1},{kw-1})][cout]")
    for (r, cc) in ((0, 0), (0, 1), (5, 7), (15, 15)):
        exp_pix = (r + kh - 1) * 16 + (cc + kw - 1) if 0 <= r + kh - 1 < 16 and 0 <= cc + kw - 1 < 16 else -1
        print(f" pixel ({r},{cc}) expect src pixel {exp_pix}: src pixel seen per cout[0:16] =", yp[0, :16, r, cc].int().tolist())
        print(f"                      src channel seen per cout[0:16] =", yc[0, :16, r, cc].int().tolist(), " cout[64:72] =", yc[0, 64:72, r, cc].int().tolist())
    ok = 0
    ref_p = torch.zeros(1, c, h, w); 
    sh = torch.nn.functional.pad(xpix, (1, 1, 1, 1))[:, :, kh:kh + 16, kw:kw + 16]
    print("   fraction of outputs with the right source pixel:", float((yp == sh).float().mean()), " right source channel:", float(((yc == xch) | (sh < 0)).float().mean()))
# chunk test: weights that copy channel ci -> cout (ci+64)%128 at the centre tap (exercises chunk B -> couts of wave_c 0)
wt = torch.zeros(c, c, 3, 3); wt[(torch.arange(c) + 64) % c, torch.arange(c), 1, 1] = 1.0
yc = run(xch, wt)
print("== channel rotate by 64 at centre tap: cout k should show channel (k+64)%128:", yc[0, :8, 3, 3].int().tolist(), yc[0, 64:72, 3, 3].int().tolist())
