"""round 6: bench-like loop (forward, backward, Adam, zero_grad; the host never waits inside a trial) repeated from the same initial state;
compares the end state across trials and, with CHECK=1, the per-step integer checksum of every gradient (taken on the device, no host
sync) to name the first tensor that differs."""
import copy
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import torch                                                                            # noqa: E402
import bench                                                                            # noqa: E402
from mas_hip import ops, optim                                                          # noqa: E402
from models import VQBASE                                                               # noqa: E402

trials, steps, batch = int(os.environ.get("TRIALS", "25")), int(os.environ.get("STEPS", "12")), int(os.environ.get("BATCH", "32"))
check = int(os.environ.get("CHECK", "0"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.set_compute_dtype(torch.bfloat16)
m = VQBASE(**bench.IMG_CFG).to(dev).train()
m.quantize.q_counter = m.quantize.q_re_end
sd0 = copy.deepcopy(m.state_dict())
x = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
names = [n for n, _ in m.named_parameters()]
params = [p for _, p in m.named_parameters()]


fwd_names, fwd_sums = [], []
if check >= 2:                        # integer checksum of every leaf module's output, in call order
    def hook(mod, inp, out, name=None):
        o = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(o) and o.is_cuda and o.dtype in (torch.float32, torch.bfloat16):
            v = o.detach().contiguous()
            v = v.view(torch.int32) if v.dtype == torch.float32 else v.view(torch.int16).to(torch.int32)
            fwd_sums.append(v.sum())
            if len(fwd_names) < 400:
                fwd_names.append(name)
    for n_, mod in m.named_modules():
        if n_:
            mod.register_forward_hook(lambda a, b, c, n_=n_: hook(a, b, c, n_))


def trial():
    m.load_state_dict(sd0)
    ops.invalidate_weight_cache()
    opt = optim.Adam(m.parameters(), lr=5e-6, betas=(0.5, 0.9))
    sums = []
    for _ in range(steps):
        rec, q = m(x)
        loss = (x - rec).abs().mean() + q
        loss.backward()
        if check:
            g = torch.stack([p.grad.view(torch.int32).sum() for p in params])
        opt.step()
        if check:
            pa = torch.stack([p.detach().view(torch.int32).sum() for p in params])
            f = torch.stack(fwd_sums) if fwd_sums else torch.zeros(1, dtype=torch.int32, device=dev)
            fwd_sums.clear()
            sums.append((f, g, pa))
        opt.zero_grad(set_to_none=True)
    end = torch.stack([p.detach().view(torch.int32).sum() for p in params])
    torch.cuda.synchronize()
    return float(loss), end.cpu(), [tuple(t.cpu() for t in s) for s in sums]


l0, e0, s0 = trial()
nbad = 0
for t in range(trials):
    l, e, s = trial()
    if l != l0 or not torch.equal(e, e0):
        nbad += 1
        msg = f"  trial {t}: final loss {l} vs {l0}; {int((e != e0).sum())} parameters differ at the end"
        done = False
        for k, (a, b) in enumerate(zip(s0, s)):
            for what, u, v, nm in (("forward output", a[0], b[0], fwd_names), ("gradient", a[1], b[1], names), ("parameter after Adam", a[2], b[2], names)):
                d = (u != v).nonzero().flatten().tolist()
                if d:
                    msg += f"; FIRST difference at step {k}: {what}: {[nm[i % len(nm)] if nm else i for i in d[:8]]} ({len(d)} of {len(u)} tensors)"
                    done = True
                    break
            if done:
                break
        print(msg)
print(f"MAS_WGRAD_STREAM={os.environ.get('MAS_WGRAD_STREAM', '1')} CHECK={check}: "
      f"{trials} trials of {steps} steps, {nbad} differ from the first")
