#!/usr/bin/env python3
"""Can an HBM-bound pass (GroupNorm backward) and an MFMA-bound kernel (weight gradient) run side by side on disjoint CU sets?
Times, at the dominant shape (128 ch @256^2 x 32): each kernel alone on the whole chip, each alone on its CU share (masked stream),
and both together -- for a few splits of the 256 CUs.  (DESIGN R3.3 item 1.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
import torch  # noqa: E402
import mas_hip  # noqa: E402
from mas_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
L = mas_hip.lib()
n, c, h = 32, 128, 256
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
da = torch.randn_like(x)
dy = torch.randn_like(x)
g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
mr, ss = ops.gn_stats(x, g, b, 32, 1e-6)
geo = (n, h, h, c, h, h, c, 3, 1, 1, 1)


def masked_stream(first, count):
    p = C.c_void_p()
    mas_hip.check(L.mas_stream_create_cu_range(first, count, C.byref(p)), "stream_create_cu_range")
    return torch.cuda.ExternalStream(p.value, device=dev)


def timed(fn_list, iters=10):
    """fn_list: [(stream, fn)]: each fn runs `iters` times on its stream; wall time of the slowest, per iteration, by events on each"""
    torch.cuda.synchronize()
    evs = []
    for st, fn in fn_list:
        with torch.cuda.stream(st):
            for _ in range(2):
                fn()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record(torch.cuda.current_stream())
    for st, fn in fn_list:
        st.wait_event(start)
    for st, fn in fn_list:
        with torch.cuda.stream(st):
            for _ in range(iters):
                fn()
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            evs.append(e)
    torch.cuda.synchronize()
    return [start.elapsed_time(e) / iters for e in evs]


main = torch.cuda.current_stream()
gnb = lambda: ops.gn_bwd(x, da, None, 32, 2, g, mr, ss)
gact = lambda: ops.gn_act(x, ss, 2)
wg = lambda: ops.conv_wgrad_raw(x, None, dy, *geo, 0, False, True)
print("whole chip, alone: gn_bwd %.3f ms | gn_act %.3f ms | wgrad %.3f ms" % (timed([(main, gnb)])[0], timed([(main, gact)])[0], timed([(main, wg)])[0]))
for gcus in (32, 64, 96, 128):
    wcus = 256 - gcus
    for lay, (g0, w0) in (("gn high", (wcus, 0)), ("gn low", (0, gcus))):
        sg, sw = masked_stream(g0, gcus), masked_stream(w0, wcus)
        mas_hip.check(L.mas_set_cu_budget(wcus), "budget")
        a = timed([(sg, gnb)])[0]
        a2 = timed([(sg, gact)])[0]
        bb = timed([(sw, wg)])[0]
        both = timed([(sg, gnb), (sw, wg)])
        mas_hip.check(L.mas_set_cu_budget(0), "budget")
        print("gn on %3d CUs (%s) / wgrad on %3d: alone gn_bwd %.3f ms (%.2f TB/s) gn_act %.3f | alone wgrad %.3f | together gn_bwd %.3f wgrad %.3f"
              % (gcus, lay, wcus, a, 5 * x.numel() * 2 / a / 1e9, a2, bb, both[0], both[1]), flush=True)
        del sg, sw
