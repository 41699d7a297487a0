#!/usr/bin/env python3
"""Shader clock and package power WHILE a kernel loop runs (turn "power budget" into a measurement).
    python tools/probes/clock_power.py            # idle, the pure-MFMA probe, the wide conv, the LDS-DMA wgrad, GroupNorm backward
Samples the amdgpu hwmon / sysfs files of card 0 every ~20 ms in a thread (sclk: freq1_input or pp_dpm_sclk's active level; power:
power1_average / power1_input, microwatts) and, where sysfs has nothing, `rocm-smi --showclocks --showpower --json` once per second."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _my_card():
    """the DRM card of HIP device 0 of THIS process (the box has many GPUs; card0 is usually somebody else's): matched by PCI address"""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); "
                            "print('%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id))"],
                           capture_output=True, text=True, timeout=300)
        want = r.stdout.strip().splitlines()[-1].lower()
    except Exception:
        return None
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.basename(os.path.realpath(card)).lower().startswith(want):
            return card
    return None


def _find():
    out = {"sclk": None, "power": None, "dpm": None}
    mine = _my_card()
    print("HIP device 0 is", mine)
    for card in ([mine] if mine else sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))):
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for k, names in (("sclk", ("freq1_input",)), ("power", ("power1_average", "power1_input"))):
                for n in names:
                    p = os.path.join(hw, n)
                    if out[k] is None and os.path.exists(p):
                        out[k] = p
        p = os.path.join(card, "pp_dpm_sclk")
        if out["dpm"] is None and os.path.exists(p):
            out["dpm"] = p
        if out["sclk"] or out["power"] or out["dpm"]:
            break
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.on, self.rows = files, True, []

    def run(self):
        use_smi = not (self.files["sclk"] or self.files["dpm"]) or not self.files["power"]
        last_smi = 0.0
        while self.on:
            mhz = watts = None
            v = _read(self.files["sclk"]) if self.files["sclk"] else None
            if v and v.isdigit():
                mhz = int(v) / 1e6
            elif self.files["dpm"]:
                for line in (_read(self.files["dpm"]) or "").splitlines():
                    if line.rstrip().endswith("*"):
                        mhz = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            v = _read(self.files["power"]) if self.files["power"] else None
            if v and v.isdigit():
                watts = int(v) / 1e6
            if use_smi and time.time() - last_smi > 1.0:
                last_smi = time.time()
                try:
                    j = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout)
                    c0 = j.get("card0", {})
                    for k, val in c0.items():
                        if mhz is None and "sclk" in k.lower() and "(" in str(val):
                            mhz = float(str(val).split("(")[1].split("Mhz")[0].split("MHz")[0])
                        if watts is None and "power" in k.lower() and "(w)" in k.lower():
                            watts = float(val)
                except Exception:
                    pass
            self.rows.append((time.time(), mhz, watts))
            time.sleep(0.02)


def measure(name, cmd, files, seconds_hint):
    s = Sampler(files)
    s.start()
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, PYTHONPATH=os.path.join(ROOT, "make-a-scene_amd")))
    t1 = time.time()
    s.on = False
    s.join()
    # the kernel loop is the tail of the process (imports and allocation come first): keep the last `seconds_hint` seconds
    rows = [r_ for r_ in s.rows if t1 - seconds_hint <= r_[0] <= t1 - 0.1]
    mhz = [r_[1] for r_ in rows if r_[1]]
    w = [r_[2] for r_ in rows if r_[2]]
    tail = [l for l in r.stdout.splitlines() if l.strip()][-2:]
    avg = lambda v: sum(v) / len(v) if v else float("nan")
    print(f"{name:34s} sclk avg {avg(mhz):7.0f} MHz (min {min(mhz) if mhz else float('nan'):.0f}, max {max(mhz) if mhz else float('nan'):.0f}, {len(mhz)} samples)   "
          f"power avg {avg(w):6.0f} W (max {max(w) if w else float('nan'):.0f})   wall {t1 - t0:.1f} s")
    for l in tail:
        print("      " + l[:200])
    sys.stdout.flush()


def main():
    files = _find()
    print("sources:", files)
    kb = [sys.executable, os.path.join(ROOT, "tools", "kbench.py")]
    measure("idle (sleep 3)", ["sleep", "3"], files, 3)
    probe = os.path.join(ROOT, "tools", "probes", "mfma_peak")
    if os.path.exists(probe):
        measure("pure MFMA loop (mfma_peak x3)", ["bash", "-c", f"{probe}; {probe}; {probe}"], files, 4)
    measure("conv3x3_wide 128->128 @256^2", kb + ["conv_fwd", "--c", "128", "--hw", "256", "--iters", "6000"], files, 3)
    measure("conv_wgrad_dma 128->128 @256^2", kb + ["wgrad", "--c", "128", "--hw", "256", "--iters", "5000"], files, 3)
    measure("gn_bwd (HBM-bound) 128 ch @256^2", kb + ["gn_bwd", "--c", "128", "--hw", "256", "--three", "1", "--iters", "6000"], files, 3)
    measure("gn_act (HBM-bound) 128 ch @256^2", kb + ["gn_act", "--c", "128", "--hw", "256", "--iters", "12000"], files, 3)


if __name__ == "__main__":
    main()
