import sys, torch
sys.path.insert(0, "make-a-scene_amd")
from mas_hip import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n in (32, 96):
    x = torch.randn(n, 128, 256, 256, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    ms = timeit(lambda: y.copy_(x))
    print(f"torch copy_ n={n}: {ms:.4f} ms  {2*x.numel()*2/ms/1e6:.1f} GB/s (1 read + 1 write)")
    g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    mr, ss = ops.gn_stats(x, g, b, 32, 1e-6)
    ms = timeit(lambda: ops.gn_act(x, ss, 2))
    print(f"gn_act n={n}: {ms:.4f} ms  {2*x.numel()*2/ms/1e6:.1f} GB/s (1 read + 1 write)")
    ms = timeit(lambda: ops.gn_stats(x, g, b, 32, 1e-6))
    print(f"gn_stats n={n}: {ms:.4f} ms  {x.numel()*2/ms/1e6:.1f} GB/s (1 read)")
    da = torch.randn_like(x)
    ms = timeit(lambda: ops.gn_bwd(x, da, None, 32, 2, g, mr, ss, path="three"))
    print(f"gn_bwd three n={n}: {ms:.4f} ms  {5*x.numel()*2/ms/1e6:.1f} GB/s (x, da twice + dx)")
    z = torch.empty_like(x)
    ms = timeit(lambda: torch.add(x, da, out=z))
    print(f"torch add n={n}: {ms:.4f} ms  {3*x.numel()*2/ms/1e6:.1f} GB/s (2 reads + 1 write)")
    del x, y, da, z
