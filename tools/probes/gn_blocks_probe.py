"""How many work-groups should the element-wise GroupNorm passes launch?  torch's own add (2 reads + 1 write, one linear sweep of
one-shot blocks) reaches 6.1 TB/s on tensors the Infinity Cache cannot hold, gn_bwd's apply pass (the same traffic) ~4.7: the apply
grid is (gx, N) with gx = MAS_GN_APPLY_BLOCKS / N, so at the default 4096 sixteen images are walked at once (96 DRAM fronts).  This
probe times gn_bwd (three launches) and gn_act with the block budget raised (fewer images in flight, thinner blocks), each setting in
its own process (the knobs are read once), against torch.add / copy_ on the same tensors.
    python tools/probes/gn_blocks_probe.py            # the sweep
    python tools/probes/gn_blocks_probe.py child      # one setting (env), used by the sweep and under rocprofv3"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "make-a-scene_amd"))

SHAPES = ((32, 128, 256), (96, 128, 256), (32, 256, 128), (32, 128, 128))


def child():
    import torch
    from mas_hip import ops
    dev = torch.device("cuda:0")

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ref = os.environ.get("GN_PROBE_TORCH", "0") == "1"
    for (n, c, hw) in SHAPES:
        x = torch.randn(n, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        da = torch.randn_like(x)
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        mr, ss = ops.gn_stats(x, g, b, 32, 1e-6)
        gb = x.numel() * 2 / 1e6
        t_act = timeit(lambda: ops.gn_act(x, ss, 2))
        t_bwd = timeit(lambda: ops.gn_bwd(x, da, None, 32, 2, g, mr, ss, path="three"))
        t_bwr = timeit(lambda: ops.gn_bwd(x, da, da, 32, 2, g, mr, ss, path="three"))
        line = (f"n={n} c={c} hw={hw}: gn_act {t_act:.4f} ms ({2 * gb / t_act:.0f} GB/s)  gn_bwd {t_bwd:.4f} ms ({5 * gb / t_bwd:.0f} GB/s)  "
                f"gn_bwd+res {t_bwr:.4f} ms ({6 * gb / t_bwr:.0f} GB/s)")
        if ref:
            z = torch.empty_like(x)
            t_add = timeit(lambda: torch.add(x, da, out=z))
            t_cp = timeit(lambda: z.copy_(x))
            line += f"  | torch add {t_add:.4f} ms ({3 * gb / t_add:.0f} GB/s) copy_ {t_cp:.4f} ms ({2 * gb / t_cp:.0f} GB/s)"
            del z
        print(line, flush=True)
        del x, da


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    first = True
    for blocks in (4096, 8192, 16384, 32768, 65536):
        env = dict(os.environ, MAS_GN_APPLY_BLOCKS=str(blocks), MAS_GN_ACT_BLOCKS=str(blocks), GN_PROBE_TORCH="1" if first else "0")
        print(f"== MAS_GN_APPLY_BLOCKS = MAS_GN_ACT_BLOCKS = {blocks}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=600)
        first = False


if __name__ == "__main__":
    main()
