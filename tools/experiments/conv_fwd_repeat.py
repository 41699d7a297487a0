"""round 6: does plain repetition of ONE launch of the general convolution kernel expose the stage-0 barrier hole (conv_fwd.hip before the
fix, MAS_HIP_LIB=<a library built from the old file>)?  The encoder's last convolution (32 x 512 x 16 x 16 -> 256, bf16 in, fp32 out) and a few
other shapes of the general kernel, REPS launches each, an integer checksum of the output per launch on the device."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
from mas_hip import ops, ACT_NONE                                                       # noqa: E402

reps = int(os.environ.get("REPS", "20000"))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
noise = os.environ.get("NOISE", "0") == "1"
big = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for (n, cin, h, cout, ks, od) in ((32, 512, 16, 256, 3, torch.float32), (8, 64, 32, 64, 3, torch.float32), (16, 256, 16, 256, 1, torch.float32),
                                   (4, 96, 24, 160, 3, torch.bfloat16)):
    x = cl(torch.randn(n, cin, h, h, generator=g).bfloat16())
    w = torch.nn.Parameter((torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(dev))
    pad = ks // 2
    sums = []
    for r in range(reps):
        if noise and r % 8 == 0:
            big.add_(1.0)                         # something else on the memory system between launches
        y = ops.conv_fwd_raw(x, None, ops.ConvWeight(w, False), None, None, n, h, h, cin, h, h, cout, ks, 1, pad, pad, ACT_NONE, False, od)
        v = y.view(torch.int32) if od == torch.float32 else y.view(torch.int16).to(torch.int32)
        sums.append(v.sum())
    t = torch.stack(sums).cpu()
    print(f"conv {n}x{cin}x{h}x{h} -> {cout} ks {ks} out {od}: {reps} launches, {int((t != t[0]).sum())} differ from the first", flush=True)
