import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "make-a-scene_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# no pretrained LPIPS / VGG16 weights exist offline: the suite checks arithmetic and layout with synthetic weights, so the "weights
# missing" error of losses/lpips.py is a warning here (tests/test_losses_host.py checks the strict default explicitly)
os.environ.setdefault("MAS_LPIPS_STRICT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _session_state_is_restored():
    """the global compute dtype (bf16 / fp32 parity mode) and the fused-statistics switch never leak from one test into the next"""
    from mas_hip import ops
    dt, on = ops.compute_dtype(), ops._stats_state["on"]
    yield
    ops.set_compute_dtype(dt)
    ops._stats_state["on"] = on
