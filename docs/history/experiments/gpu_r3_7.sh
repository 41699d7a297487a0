#!/bin/bash
# round-3 GPU call 9: materialised GroupNorm(+SiLU) output in training (mas_gn_act) vs fused loaders: correctness + same-box A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_7; mkdir -p $O
cd $R
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
echo "== pytest, fused loaders everywhere (MAS_GN_MATERIALIZE=0): model-level parity"
MAS_GN_MATERIALIZE=0 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_losses.py -m gpu -q 2>&1 | tail -3
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: %s  frac %.4f  encoder %.3f ms  loss %.5f" % (d["value"], d["ms_per_step"], {k:(v["launches"], v["avg_launch_ms"]) for k,v in r.items()}, d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"], d["final_loss"]))
except Exception as e: print("  (no result)", e)
P
}
i=0
for v in "MAS_GN_MATERIALIZE=1" "MAS_GN_MATERIALIZE=0" "MAS_GN_MATERIALIZE=1" "MAS_GN_MATERIALIZE=0"; do
  i=$((i+1)); n=$O/bench_$i.json
  echo "== bench $i [$v]"
  env $v $B > $n 2> $n.err
  short $n
done
echo "== kbench gn_act-equivalent passes"
KB="timeout 120 python tools/kbench.py"
python - <<'P'
import sys, os, torch
sys.path.insert(0, "make-a-scene_amd")
from mas_hip import ops
dev = torch.device("cuda:0")
for (n, c, h) in ((32, 128, 256), (32, 128, 128), (32, 256, 64), (32, 512, 32), (32, 512, 16)):
    x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    ss = torch.randn(n, c, 2, device=dev)
    for _ in range(3): ops.gn_act(x, ss, 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gn_act(x, ss, 2)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("gn_act n=%d c=%d hw=%d: %.4f ms  %.1f GB/s (read + write)" % (n, c, h, ms, 2 * x.numel() * 2 / ms / 1e6))
P
