"""LayerNorm forward at the transformer's shape (12288 x 1024): fp32 -> bf16 (pre-LN) and bf16 -> fp32 + residual (sandwich LN), us per launch.
   MAS_LN_FWD_BLOCKS_PER_CU=4|8 python tools/probes/ln_probe.py"""
import sys, torch
sys.path.insert(0, "make-a-scene_amd")
from mas_hip import ops
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
rows, d = 12288, 1024
w, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
x32 = torch.randn(8, 1536, d, device="cuda"); xb = x32.bfloat16(); res = torch.randn(8, 1536, d, device="cuda")
with torch.no_grad():
    t1 = timeit(lambda: ops.layer_norm(x32, w, b, 1e-5, None, torch.bfloat16))
    t2 = timeit(lambda: ops.layer_norm(xb, w, b, 1e-5, res, torch.float32))
print(f"layernorm fwd f32->bf16 {t1:.1f} us ({75.5e6/t1/1e6:.2f} TB/s)   bf16->f32 + residual {t2:.1f} us ({125.8e6/t2/1e6:.2f} TB/s)")
