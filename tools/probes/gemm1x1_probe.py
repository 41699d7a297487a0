"""What would the library GEMM do on the 1x1-convolution shapes of the VQ-IMG step?  (pixels x Cin) @ (Cin x Cout), bf16, fp32 accumulate.
   python tools/probes/gemm1x1_probe.py"""
import torch


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    for (px, cin, cout) in ((8192, 512, 1536), (8192, 512, 512), (8192, 1536, 512), (524288, 256, 128), (524288, 128, 256), (131072, 512, 256),
                            (131072, 256, 512), (32768, 256, 512), (32768, 512, 256), (8192, 256, 256)):
        a = torch.randn(px, cin, device=dev).bfloat16()
        w = torch.randn(cout, cin, device=dev).bfloat16()
        b = torch.randn(cout, device=dev).bfloat16()
        dy = torch.randn(px, cout, device=dev).bfloat16()
        r = torch.randn(px, cout, device=dev).bfloat16()
        t_f = timeit(lambda: torch.addmm(b, a, w.t()))
        t_r = timeit(lambda: torch.addmm(r, a, w.t()))
        t_d = timeit(lambda: torch.mm(dy, w))
        t_w = timeit(lambda: torch.mm(dy.t(), a))
        fl = 2.0 * px * cin * cout
        print(f"px={px:7d} {cin:4d}->{cout:4d}: fwd+bias {t_f:7.1f} us ({fl/t_f/1e6:5.0f} TF)  fwd+res {t_r:7.1f}  dgrad {t_d:7.1f} us  wgrad {t_w:7.1f} us")


if __name__ == "__main__":
    main()
