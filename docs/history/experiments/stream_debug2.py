import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch
from mas_hip import ops
dev = torch.device("cuda:0")
n, c, h, w = 1, 128, 16, 16
def run(x, wt):
    wp = ops.pack_conv_weight(wt.to(dev), False, torch.bfloat16)
    xd = x.bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd_raw(xd, None, wp, None, None, n, h, w, c, h, w, c, 3, 1, 1, 1, 0, False, torch.bfloat16)
    torch.cuda.synchronize()
    return y.float().cpu()
xch = torch.arange(c).float()[None, :, None, None].expand(1, c, h, w).contiguous()
wt = torch.zeros(c, c, 3, 3); wt[torch.arange(c), torch.arange(c), 1, 1] = 1.0
print("identity:", run(xch, wt)[0, :, 3, 3].int().tolist())
wt = torch.zeros(c, c, 3, 3); wt[:, 5, 1, 1] = 1.0
print("all couts copy channel 5:", run(xch, wt)[0, :, 3, 3].int().tolist())
wt = torch.zeros(c, c, 3, 3); wt[:, 70, 1, 1] = 1.0
print("all couts copy channel 70:", run(xch, wt)[0, :, 3, 3].int().tolist())
wt = torch.zeros(c, c, 3, 3); wt[7, :, 1, 1] = 1.0
print("cout 7 sums all channels (8128):", run(xch, wt)[0, :, 3, 3].int().tolist())
x1 = torch.zeros(1, c, h, w); x1[0, 9] = 1.0
wt = torch.zeros(c, c, 3, 3); wt[:, :, 1, 1] = (torch.arange(c)[:, None] * 1.0 + 0 * torch.arange(c)[None, :])   # W[co][ci] = co
print("W[co][ci]=co, x=e_9 -> y[co]=co:", run(x1, wt)[0, :, 3, 3].int().tolist())
wt = torch.zeros(c, c, 3, 3); wt[:, :, 1, 1] = (0 * torch.arange(c)[:, None] + 1.0 * torch.arange(c)[None, :])   # W[co][ci] = ci
for ch in (0, 9, 40, 70):
    x1 = torch.zeros(1, c, h, w); x1[0, ch] = 1.0
    print(f"W[co][ci]=ci, x=e_{ch} -> y[co]={ch}:", run(x1, wt)[0, :, 3, 3].int().tolist()[:20], "...")
