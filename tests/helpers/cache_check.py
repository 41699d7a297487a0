"""Child process of tests/test_gpu_parity_r3.py::test_weight_cache_check_flags_a_write_through_data (MAS_WEIGHT_CACHE_CHECK is read when
mas_hip.ops is imported, so the knob needs its own process).  Prints one line per stage; exit code 0 when the three stages behaved."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "make-a-scene_amd"))
from mas_hip import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(128, 64, 3, 3, device=dev) * 0.05)
    x = torch.randn(2, 64, 32, 32, device=dev)
    y0 = ops.norm_act_conv(x, w, None).float()
    y1 = ops.norm_act_conv(x, w, None).float()                 # cache hit, checksum unchanged
    print("hit ok", bool(torch.equal(y0, y1)))
    w.data.mul_(2.0)                                           # a write the stamp cannot see
    try:
        ops.norm_act_conv(x, w, None)
        print("stale hit NOT flagged")
        return 1
    except RuntimeError as e:
        print("flagged:", str(e)[:60])
    ops.invalidate_weight_cache()
    y2 = ops.norm_act_conv(x, w, None).float()
    ok = bool(torch.allclose(y2, 2.0 * y0, rtol=2e-2, atol=2e-2))
    print("after invalidate ok", ok)
    return 0 if ok and torch.equal(y0, y1) else 1


if __name__ == "__main__":
    sys.exit(main())
