#!/usr/bin/env python3
"""Micro-benchmark of single libmas_hip kernels on synthetic shapes (HIP-event timed).
    python tools/kbench.py conv_fwd|dgrad|wgrad|gn_stats|gn_bwd|gn_act|vq [--n 32 --c 128 --hw 256 --act 2 --iters 20]
Prints achieved TFLOP/s / GB/s per launch; run under rocprofv3 for counters."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch  # noqa: E402
from mas_hip import ops  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind")
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--c", type=int, default=128)
    ap.add_argument("--co", type=int, default=0)
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--ks", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1, help="2 = the Downsample geometry (pad 0/1, conv_fwd only)")
    ap.add_argument("--act", type=int, default=0)
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--ups", type=int, default=0, help="conv_fwd: 1 = Upsample's convolution (nearest x2 folded in: --hw is the INPUT map, the output is 2x)")
    ap.add_argument("--stats", type=int, default=0, help="conv_fwd: 1 = request the fused GroupNorm statistics epilogue (the training forward's variant)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--three", type=int, default=-1, help="gn_bwd: 1 = the three-launch path, 2 = the library default (small-map kernel where it applies), -1 = both")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--zero", type=int, default=0, help="1 = all-zero activations and weights (no operand toggling: the power-bound clock give-back)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    co = a.co or a.c
    n, c, h = a.n, a.c, a.hw
    x = torch.randn(n, c, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    esz = x.element_size()
    if a.zero:
        x.zero_()
    if a.kind in ("conv_fwd", "dgrad", "wgrad"):
        w = (torch.randn(co, c, a.ks, a.ks, device=dev) / (c * a.ks * a.ks) ** 0.5)
        if a.zero:
            w.zero_()
        b = torch.randn(co, device=dev) * 0.1
        ss = torch.randn(n, c, 2, device=dev) if a.act else None
        p = a.ks // 2
        flops = 2.0 * a.ks * a.ks * c * co * n * h * h
        if a.kind == "wgrad":
            ho = 2 * h if a.ups else h             # --ups 1: Upsample's convolution (x is the low-resolution map, dy the x2 one)
            dy = torch.randn(n, co, ho, ho, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
            fn = lambda: ops.conv_wgrad_raw(x, ss, dy, n, h, h, c, ho, ho, co, a.ks, 1, p, p, a.act, bool(a.ups), True)
            byt = (x.numel() + dy.numel()) * esz
            if a.ups:
                flops *= 4.0
        else:
            wp = ops.ConvWeight(torch.nn.Parameter(w), a.kind == "dgrad")   # packed (once: the Parameter is cached) in the layout the library prefers
            res = torch.randn(n, co, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last) if a.res else None
            ho = (2 * h if a.ups else h) if a.stride == 1 else h // 2
            pt = p if a.stride == 1 else 0
            if a.ups:
                flops *= 4.0                         # counted as the reference runs it: 9 taps at the output resolution
                res = None
            if a.stride != 1:
                flops /= 4.0
                res = None
            fn = lambda: ops.conv_fwd_raw(x, ss, wp, b, res, n, h, h, c, ho, ho, co, a.ks, a.stride, pt, pt, a.act, bool(a.ups), dt, want_stats=bool(a.stats))
            byt = (x.numel() + n * co * ho * ho * (2 if res is not None else 1)) * esz
        ms = timeit(fn, a.iters)
        print(f"{a.kind} n={n} c={c}->{co} hw={h} ks={a.ks} act={a.act} {a.dtype}: {ms:.4f} ms  {flops/ms/1e9:.1f} TFLOP/s  {byt/ms/1e6:.1f} GB/s(alg)")
    elif a.kind == "gn_stats":
        g, bta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        ms = timeit(lambda: ops.gn_stats(x, g, bta, 32, 1e-6), a.iters)
        print(f"gn_stats n={n} c={c} hw={h}: {ms:.4f} ms  {x.numel()*esz/ms/1e6:.1f} GB/s")
    elif a.kind == "gn_bwd":
        g, bta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        mr, ss = ops.gn_stats(x, g, bta, 32, 1e-6)
        da = torch.randn_like(x)
        dres = torch.randn_like(x) if a.res else None
        alg = (3 + (1 if a.res else 0)) * x.numel() * esz         # x, da (, dres) read once, dx written once
        for path in ((None, "three") if a.three < 0 else (("three",) if a.three == 1 else (None,))):
            ms = timeit(lambda: ops.gn_bwd(x, da, dres, 32, a.act or 2, g, mr, ss, path=path), a.iters)
            name = {None: "default: " + ops.last_kernel(), "three": "three launches"}[path]
            print(f"gn_bwd[{name}] n={n} c={c} hw={h} res={a.res}: {ms:.4f} ms  "
                  f"{alg/ms/1e6:.1f} GB/s algorithmic (x + da{' + dres' if a.res else ''} + dx once)")
    elif a.kind == "gn_fwd":
        g, bta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        def three():
            mr, ss = ops.gn_stats(x, g, bta, 32, 1e-6)
            return ops.gn_act(x, ss, a.act or 2)
        ms3 = timeit(three, a.iters)
        print(f"gn_fwd[stats + finalize + act: three launches] n={n} c={c} hw={h}: {ms3:.4f} ms")
        if ops.gn_small_ok(x, 32):
            ms1 = timeit(lambda: ops.gn_stats_act(x, g, bta, 32, 1e-6, a.act or 2), a.iters)
            print(f"gn_fwd[{ops.last_kernel()}: one launch] n={n} c={c} hw={h}: {ms1:.4f} ms")
    elif a.kind == "gn_act":
        g, bta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        mr, ss = ops.gn_stats(x, g, bta, 32, 1e-6)
        ms = timeit(lambda: ops.gn_act(x, ss, a.act or 2), a.iters)
        print(f"gn_act n={n} c={c} hw={h}: {ms:.4f} ms  {2*x.numel()*esz/ms/1e6:.1f} GB/s (1 read + 1 write)")
    elif a.kind == "attn":
        b, h, sq, hd = a.n, 16, a.hw if a.hw != 256 else 1536, 64
        qkv = torch.randn(b, sq, 3 * h * hd, device=dev).to(dt).requires_grad_(True)
        fl = 4.0 * b * h * sq * sq * hd / 2          # causal: half of the full S x S products
        ms = timeit(lambda: ops.causal_attention(qkv.detach(), h), a.iters)
        print(f"attn fwd B={b} H={h} S={sq} hd={hd} {a.dtype}: {ms:.4f} ms  {fl/ms/1e9:.1f} TFLOP/s (causal FLOPs)")
        import ctypes
        from mas_hip import lib
        tr = getattr(lib(), "mas_fa_trace", None)           # only in a -DFA_TRACE variant build (tools/build_file_variant.sh)
        if tr is not None:
            ops.causal_attention(qkv.detach(), h); torch.cuda.synchronize()
            buf = (ctypes.c_longlong * 32)()
            tr.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
            tr(buf)
            t = list(buf)
            names = ["prologue (Q, first DMA)", "DMA issue", "K reads + score MFMAs", "mask + softmax", "cvt + V reads + PV MFMAs", "DMA wait", "barrier", "epilogue"]
            nt = max(int(t[8]), 1)
            print(f"attn fwd phases of wave 3 of the heaviest work-group ({nt} key tiles; us total / ns per tile): "
                  + "; ".join(f"{nm} {t[i]/100:.2f}" + (f" / {t[i]*10/nt:.0f}" if 1 <= i <= 6 else "") for i, nm in enumerate(names)))
            print(f"  that work-group ran {(t[10]-t[9])/100:.2f} us; the lightest work-group of the same head started {(t[11]-t[9])/100:.2f} us after it and ran {(t[12]-t[11])/100:.2f} us")
        go = torch.randn(b, sq, h * hd, device=dev).to(dt)
        def fb():
            qkv.grad = None
            ops.causal_attention(qkv, h).backward(go)
        ms2 = timeit(fb, a.iters)
        print(f"attn fwd+bwd: {ms2:.4f} ms  {3.5*fl/ms2/1e9:.1f} TFLOP/s (fwd 1x + bwd 2.5x causal FLOPs)")
    elif a.kind == "sp_attn":
        cc = c if c != 128 else 512
        hh = h if h != 256 else 16
        qkv = torch.randn(n, 3 * cc, hh, hh, device=dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        fl = 4.0 * n * (hh * hh) ** 2 * cc
        ms = timeit(lambda: ops.spatial_attention(qkv.detach(), cc), a.iters)
        print(f"sp_attn fwd n={n} S={hh*hh} C={cc}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
        go = torch.randn(n, cc, hh, hh, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        def fb():
            qkv.grad = None
            ops.spatial_attention(qkv, cc).backward(go)
        ms2 = timeit(fb, a.iters)
        print(f"sp_attn fwd+bwd: {ms2*1e3:.1f} us  {3.5*fl/ms2/1e9:.1f} TFLOP/s (backward = 2.5x forward FLOPs)")
        import ctypes
        from mas_hip import lib
        tr = getattr(lib(), "mas_sp_trace", None)           # only in a -DSP_TRACE variant build
        if tr is not None:
            ops.spatial_attention(qkv.detach(), cc); torch.cuda.synchronize()
            buf = (ctypes.c_longlong * 32)()
            tr.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
            tr(buf)
            t = list(buf)
            names = ["stage Q + prod QK", "issue V", "softmax", "apply PV", "store"]
            print("sp_attn forward phases (us, work-group 9): " + ", ".join(f"{nm} {(t[i+1]-t[i])/100:.2f}" for i, nm in enumerate(names)))
            print("  apply tile ends (us after apply start): " + " ".join(f"{(t[6+i]-t[3])/100:.2f}" for i in range(8)))
            print("  prod (us after kernel start): loads issued %.2f, block rows staged %.2f, steps done " % ((t[16]-t[0])/100, (t[17]-t[0])/100)
                  + " ".join(f"{(t[18+i]-t[0])/100:.2f}" for i in range(8)))
    elif a.kind == "vq":
        z = torch.randn(n, 256, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
        cb = torch.randn(8192, 256, device=dev)
        ms = timeit(lambda: ops.vq_lookup(z, cb, 0.25), a.iters)
        print(f"vq M={n*256} K=8192 D=256: {ms:.4f} ms  {2.0*n*256*8192*256/ms/1e9:.1f} TFLOP/s(fp32)")


if __name__ == "__main__":
    main()
