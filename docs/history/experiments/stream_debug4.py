import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
from mas_hip import ops
dev = torch.device("cuda:0")
n, c, h, w = 1, 128, 32, 32
wt = torch.zeros(c, c, 3, 3)
for tap in range(9):
    wt[:, :, tap // 3, tap % 3] = torch.arange(c)[:, None].float() + 0.0 * torch.arange(c)[None, :]      # W[co][ci][tap] = co
wp = ops.pack_conv_weight(wt.to(dev), False, torch.bfloat16)
x = torch.arange(c).float()[None, :, None, None].expand(n, c, h, w).contiguous()
xd = x.bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
res = torch.zeros(n, c, h, w, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
y = ops.conv_fwd_raw(xd, None, wp, None, res, n, h, w, c, h, w, c, 3, 1, 1, 1, 0, False, torch.bfloat16)
torch.cuda.synchronize()
raw = res.cpu().permute(0, 2, 3, 1).reshape(-1)
wl = raw[:2 * 128 * 64].float().view(2, 128, 64)
exp = wp.float().cpu().view(2, 9, 128, 64)[0, 4:6]
print("stage 2 weights (steps 4,5): rows equal per step:", [(wl[s] == exp[s]).all(1).sum().item() for s in range(2)])
print(" step 4 first elem rows 0..40:", wl[0, :40, 0].int().tolist())
pt = raw[2 * 128 * 64: 2 * 128 * 64 + 2 * 41 * 512].float().view(2, 41 * 8, 64)       # [buf][pixel][64 ch, slots swizzled]
for buf in range(2):
    print(f" patch buf {buf}: pixel 19 (row 1, col 1 = image (0,0)) slot contents (first elem of each 16-B slot):", pt[buf, 19, ::8].int().tolist(),
          " pixel 0 (padding):", pt[buf, 0, ::8].int().tolist(), " pixel 40:", pt[buf, 40, ::8].int().tolist())
print("y[0,:,3,3] =", y[0, :, 3, 3].float().cpu().int().tolist()[:40])
