"""Does GroupNorm backward gain from walking the batch in image groups that fit the 256 MiB Infinity Cache (reduce -> finalize -> apply
per group, so that the apply pass re-reads x / da from the cache)?  Emulated with ops.gn_bwd on batch slices (contiguous in NHWC).
   python tools/probes/gn_chunk_probe.py"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "make-a-scene_amd"))
from mas_hip import ops  # noqa: E402


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


def main():
    dev = torch.device("cuda:0")
    for (c, hw) in ((128, 256), (128, 128), (256, 128)):
        n = 32
        x = torch.randn(n, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        da = torch.randn_like(x)
        dres = torch.randn_like(x)
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        mr, ss = ops.gn_stats(x, g, b, 32, 1e-6)
        for res in (None, dres):
            line = [f"c={c} hw={hw} res={'y' if res is not None else 'n'}:"]
            for ch in ((32,) if len(sys.argv) > 1 and sys.argv[1] == 'full' else (32, 16, 8, 4, 2)):
                def run():
                    for i in range(0, n, ch):
                        ops.gn_bwd(x[i:i + ch], da[i:i + ch], None if res is None else res[i:i + ch], 32, 2, g, mr[i:i + ch], ss[i:i + ch])
                line.append(f"chunk {ch}: {timeit(run):.4f} ms")
            print("  ".join(line))


if __name__ == "__main__":
    main()
