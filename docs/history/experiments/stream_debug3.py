import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
from mas_hip import ops
dev = torch.device("cuda:0")
n, c, h, w = 1, 128, 32, 32          # y has 32*32*128*2 = 256 KiB: room for the dump
wt = torch.zeros(c, c, 3, 3)
for tap in range(9):
    wt[:, :, tap // 3, tap % 3] = torch.arange(c)[:, None].float() + 0.0 * torch.arange(c)[None, :]      # W[co][ci][tap] = co
wp = ops.pack_conv_weight(wt.to(dev), False, torch.bfloat16)
img = wp.view(torch.int16).cpu().view(2, 9, 128, 64)                  # [chunk][tap][row][64 bf16]
print("packed image rows (chunk 0, tap 0), first element of rows 0..9:", wp.float().cpu().view(2, 9, 128, 64)[0, 0, :10, 0].tolist())
x = torch.zeros(n, c, h, w); 
xd = x.bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
y = ops.conv_fwd_raw(xd, None, wp, None, None, n, h, w, c, h, w, c, 3, 1, 1, 1, 0, False, torch.bfloat16)
torch.cuda.synchronize()
raw = y.cpu().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1)       # NHWC flat
wl = raw[:2 * 128 * 64].float().view(2, 128, 64)                       # LDS weight stage 0: [step][row][64]
print("LDS weight stage, step 0, first element of rows 0..40:", wl[0, :40, 0].int().tolist())
print("LDS weight stage, step 0, row 5 :", wl[0, 5, :16].int().tolist())
print("LDS weight stage, step 1, first element of rows 0..40:", wl[1, :40, 0].int().tolist())
exp = wp.float().cpu().view(2, 9, 128, 64)[0, :2]
print("matches the packed image steps 0,1:", bool((wl == exp).all()), " rows equal per step:", [(wl[s] == exp[s]).all(1).sum().item() for s in range(2)])
