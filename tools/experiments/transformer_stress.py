"""round 6: the transformer step of bench.py (MakeAScene 24L/1024d, batch 8, bf16 autocast, Adam) repeated from the same state; integer
checksums (taken on the device, no host wait inside a trial) of every module output, every gradient and every parameter after Adam per step.
Prints the first thing that differs from the first trial.  The Linear layers are library GEMMs: a difference first seen at a Linear output
is the library's, one first seen at attention / LayerNorm / GELU / embedding outputs is this repo's."""
import copy
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops, optim                                                          # noqa: E402
from models.transformer import MakeAScene                                               # noqa: E402

trials, steps, batch = int(os.environ.get("TRIALS", "40")), int(os.environ.get("STEPS", "6")), int(os.environ.get("BATCH", "8"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MakeAScene(**bench.TR_CFG).to(dev).train()
sd0 = copy.deepcopy(m.state_dict())
g = torch.Generator().manual_seed(4321)
text = torch.randint(1, 49408, (batch, 256), generator=g)
text[:, 200:] = 0
text = text.to(dev)
seg_tok = torch.randint(0, 256, (batch, 256), generator=g).to(dev)
img_tok = torch.randint(0, 8192, (batch, 1024), generator=g).to(dev)
names = [n for n, _ in m.named_parameters()]
params = [p for _, p in m.named_parameters()]
fwd_names, fwd_sums = [], []


def isum(t):
    v = t.detach().contiguous()
    return (v.view(torch.int32) if v.dtype == torch.float32 else v.view(torch.int16).to(torch.int32)).sum()


def hook(mod, inp, out, name=None):
    o = out[0] if isinstance(out, (tuple, list)) else out
    if torch.is_tensor(o) and o.is_cuda and o.dtype in (torch.float32, torch.bfloat16):
        fwd_sums.append(isum(o))
        if len(fwd_names) < 2000:
            fwd_names.append(f"{name} ({type(mod).__name__})")


for n_, mod in m.named_modules():
    if n_:
        mod.register_forward_hook(lambda a, b, c, n_=n_: hook(a, b, c, n_))


def trial():
    m.load_state_dict(sd0)
    ops.invalidate_weight_cache()
    opt = optim.Adam(m.parameters(), lr=1e-4)
    sums = []
    for _ in range(steps):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = m(text, seg_tok, img_tok)
        loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), img_tok.reshape(-1))
        loss.backward()
        gs = torch.stack([isum(p.grad) if p.grad is not None else torch.zeros((), dtype=torch.int32, device=dev) for p in params])
        opt.step()
        pa = torch.stack([isum(p) for p in params])
        f = torch.stack(fwd_sums)
        fwd_sums.clear()
        sums.append((f, gs, pa))
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    return float(loss), [tuple(t.cpu() for t in s) for s in sums]


l0, s0 = trial()
nbad = 0
for t in range(trials):
    l, s = trial()
    done = False
    for k, (a, b) in enumerate(zip(s0, s)):
        for what, u, v, nm in (("forward output", a[0], b[0], fwd_names), ("gradient", a[1], b[1], names), ("parameter after Adam", a[2], b[2], names)):
            d = (u != v).nonzero().flatten().tolist()
            if d:
                nbad += 1
                print(f"  trial {t}: FIRST difference at step {k}: {what}: {[nm[i % len(nm)] for i in d[:4]]} ({len(d)} of {len(u)} tensors)")
                done = True
                break
        if done:
            break
print(f"transformer step, batch {batch}: {trials} trials of {steps} steps, {nbad} differ from the first")
