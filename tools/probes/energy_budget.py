#!/usr/bin/env python3
"""Energy budget of the convolution kernels (VERDICT r4 "next" #1): shader clock and package power over the steady state of >= 4 s
loops of (a) the matrix pipe alone, fed constant / random operands from registers / random operands through ds_read_b128 at the
wide kernel's 0.75 reads per MFMA and at 0.5 (tools/probes/mfma_power.hip), (b) the wide convolution kernel and its ablation builds
(-DW_ABL_NOPATCH: no activation DMA, -DW_ABL_NOW: no weight DMA, -DW_ABL_NOEPI: no epilogue, all three = MFMAs + fragment reads
only, -DW_ABL_HALF_A: 0.5 fragment reads per MFMA), on random and on all-zero inputs, (c) the LDS-DMA weight gradient.
Prints one row per load: ms per launch, TFLOP/s, clock, watts, JOULES PER LAUNCH (P x t) and pJ per FLOP.
    python tools/probes/energy_budget.py [seconds per load, default 5]
The ablation libraries are looked up under make-a-scene_amd/csrc/build/variants/ (tools/experiments/gpu_r5_1.sh builds them)."""
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from clock_power import Sampler, _find  # noqa: E402


def measure(name, cmd, files, tail_s, env=None, flops=None):
    s = Sampler(files)
    s.start()
    t0 = time.time()
    e = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "make-a-scene_amd"))
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
    t1 = time.time()
    s.on = False
    s.join()
    rows = [q for q in s.rows if t1 - tail_s <= q[0] <= t1 - 0.15]
    mhz = [q[1] for q in rows if q[1]]
    w = [q[2] for q in rows if q[2]]
    avg = lambda v: sum(v) / len(v) if v else float("nan")
    out = [l for l in r.stdout.splitlines() if l.strip()]
    line = out[-1] if out else (r.stderr.strip().splitlines() or ["(no output)"])[-1]
    m = re.search(r"([0-9.]+) ms", line)
    tf = re.search(r"([0-9.]+) TFLOP/s", line)
    ms = float(m.group(1)) if m else float("nan")
    tfs = float(tf.group(1)) if tf else float("nan")
    joule = avg(w) * ms * 1e-3
    pj = avg(w) / (tfs * 1e12) * 1e12 if tfs == tfs and tfs > 0 else float("nan")
    print(f"{name:58s} {ms:8.4f} ms {tfs:7.1f} TF/s  sclk {avg(mhz):5.0f} MHz (min {min(mhz) if mhz else 0:.0f})  "
          f"power {avg(w):5.0f} W (max {max(w) if w else 0:.0f}, {len(w)} samples)  {joule:6.3f} J/launch  {pj:5.2f} pJ/FLOP   wall {t1 - t0:.1f} s")
    print("      " + line[:230])
    sys.stdout.flush()
    return ms, tfs, avg(mhz), avg(w)


def main_up2(sec):
    """joules per launch of the Upsample convolution (128 -> 128, 128^2 -> 256^2, batch 32) in its sub-pixel form against the 3x3 kernel
    with the x2 folded into the addresses (MAS_CONV_UP2=0): fewer FLOPs are fewer joules, and under the cap joules are time"""
    files = _find()
    print("sources:", files)
    kb = [sys.executable, os.path.join(ROOT, "tools", "kbench.py")]
    for c, hw in ((128, 128), (256, 64), (512, 32)):
        base = kb + ["conv_fwd", "--c", str(c), "--hw", str(hw), "--ups", "1", "--stats", "1"]
        measure(f"Upsample conv {c}->{c} @{hw}->{2*hw}: sub-pixel (conv_up2)", base + ["--iters", str(int(sec / 0.3e-3))], files, sec - 1.5, env={"MAS_CONV_UP2": "1"})
        measure(f"Upsample conv {c}->{c} @{hw}->{2*hw}: 3x3 at the high resolution", base + ["--iters", str(int(sec / 0.5e-3))], files, sec - 1.5, env={"MAS_CONV_UP2": "0"})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "up2":
        return main_up2(float(sys.argv[2]) if len(sys.argv) > 2 else 5.0)
    sec = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    files = _find()
    print("sources:", files)
    measure("idle (sleep 3)", ["sleep", "3"], files, 2.5)
    probe = os.path.join(HERE, "mfma_power")
    for mode, tile, wg, what in (("const", "42", "2", "MFMA only, constant operands, 2 waves/SIMD"),
                                 ("reg", "42", "2", "MFMA only, RANDOM operands from registers, 2 waves/SIMD"),
                                 ("reg", "42", "1", "MFMA only, RANDOM operands from registers, 1 wave/SIMD"),
                                 ("reg", "44", "1", "MFMA only, RANDOM operands, 4x4 tile, 1 wave/SIMD"),
                                 ("lds", "42", "2", "MFMA + 0.75 ds_read_b128/MFMA (random), 2 waves/SIMD"),
                                 ("lds", "44", "1", "MFMA + 0.5 ds_read_b128/MFMA (random), 4x4, 1 wave/SIMD")):
        measure(what, [probe, mode, tile, wg, str(sec)], files, sec - 1.5)
    kb = [sys.executable, os.path.join(ROOT, "tools", "kbench.py")]
    iters = str(int(sec / 0.5e-3))
    var = os.path.join(ROOT, "make-a-scene_amd", "csrc", "build", "variants")
    conv = kb + ["conv_fwd", "--c", "128", "--hw", "256", "--iters", iters]
    measure("conv3x3_wide 128->128 @256^2 (shipped), random input", conv, files, sec - 1.5)
    measure("conv3x3_wide (shipped), ALL-ZERO input and weights", conv + ["--zero", "1"], files, sec - 1.5)
    for v, what in (("wabl_nopatch", "wide -DW_ABL_NOPATCH (no activation DMA)"), ("wabl_now", "wide -DW_ABL_NOW (no weight DMA)"),
                    ("wabl_noepi", "wide -DW_ABL_NOEPI (no epilogue / stores)"), ("wabl_mfma", "wide, all three: MFMAs + fragment reads only"),
                    ("wabl_halfa", "wide -DW_ABL_HALF_A (0.5 fragment reads per MFMA)"),
                    ("wabl_mfma_halfa", "wide, MFMAs + HALF the weight fragment reads only")):
        so = os.path.join(var, v + ".so")
        if os.path.exists(so):
            measure(what, conv, files, sec - 1.5, env={"MAS_HIP_LIB": so})
    measure("conv3x3_wide <res,stats> (the forward's variant)", conv + ["--res", "1", "--stats", "1"], files, sec - 1.5)
    measure("conv_wgrad_dma 128->128 @256^2", kb + ["wgrad", "--c", "128", "--hw", "256", "--iters", iters], files, sec - 1.5)
    measure("gn_act (HBM-bound) 128 ch @256^2", kb + ["gn_act", "--c", "128", "--hw", "256", "--iters", str(int(sec / 0.2e-3))], files, sec - 1.5)


if __name__ == "__main__":
    main()
