"""round 6: which part of the N>1 path (mas_hip.dp.GradReducer at world size 1, MAS_BENCH_FORCE_DDP=1) loses the side-stream
weight-gradient gain?  Runs bench.py's own step with one part of the reducer removed (MODE env):
  full    unchanged
  pgonly  process group initialised, no reducer at all
  nohook  hooks return at once; every bucket is flattened and reduced in finish()
  noar    buckets flattened by the hooks, no all_reduce call
  nocat   hooks count, buckets are neither flattened nor reduced (.grad left as autograd produced it)
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "make-a-scene_amd"))
import bench                                                                            # noqa: E402
from mas_hip import dp                                                                  # noqa: E402

mode = os.environ.get("MODE", "full")
late = os.environ.get("LATE_QUEUES")          # "import": set GPU_MAX_HW_QUEUES after `import torch`; "avail": after torch.cuda.is_available() too
if late:
    import torch
    if late == "avail":
        torch.cuda.is_available(), torch.cuda.device_count()
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


class _Done:
    def wait(self):
        return True


if mode == "pgonly":
    bench._wrap_dp = lambda model, args, ddp, local_rank: (model, None)
elif mode == "nohook":
    dp.GradReducer._on_grad = lambda self, p: None
elif mode == "noar":
    import torch.distributed as dist
    _real = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: _Done() if k.get("async_op") else _real(t, *a, **k)
elif mode == "nocat":
    def _launch(self, b):
        b.work = _Done()
        b.launched = True
    dp.GradReducer._launch = _launch
bench.main()
