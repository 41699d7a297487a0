"""Outputs of the stream convolution kernel on small maps for the tile height selected by the calling environment
(MAS_CONV_STREAM_TH8 = 1: 8-row tiles when there are fewer 16-row tiles than CUs; 0: always 16-row tiles), saved to argv[1].
tests/test_gpu_stream_th8.py runs it under both settings and compares the tensors bit for bit: the two tilings walk the same
(tap, channel) order per output element, so they must agree exactly."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from mas_hip import ops  # noqa: E402
from th8_spec import CASES, make_case  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    for case in CASES:
        name, n, cin, cout, h, w, res, ups = case
        x, wt, b, r = make_case(case)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        rd = r.to(dev).contiguous(memory_format=torch.channels_last) if r is not None else None
        ho, wo = (2 * h, 2 * w) if ups else (h, w)
        y = ops.conv_fwd_raw(xd, None, ops.ConvWeight(wt.to(dev), False), b.to(dev), rd, n, h, w, cin, ho, wo, cout, 3, 1, 1, 1, 0, ups,
                             torch.bfloat16)
        assert ops.last_kernel() == "conv3x3_stream", (name, ops.last_kernel())
        out[name] = y.cpu()
    torch.save(out, sys.argv[1])


if __name__ == "__main__":
    main()
