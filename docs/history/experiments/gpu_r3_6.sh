#!/bin/bash
# round-3 GPU call 8: fused GroupNorm statistics on packed fp32 math (pre-rounding values, branch-free hot path): A/B on the step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_6; mkdir -p $O
cd $R
echo "== pytest (wide + model + parity)"
timeout 1500 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
echo "== pytest with fused statistics ON (model-level parity)"
MAS_FUSED_GN_STATS=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -m gpu -q -x 2>&1 | tail -3
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: plain %.4f ms gn_silu %.4f ms  frac %.4f  encoder %.3f ms  loss %.5f" % (d["value"], d["ms_per_step"], r["plain"]["avg_launch_ms"], r["gn_silu"]["avg_launch_ms"], d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"], d["final_loss"]))
except Exception as e: print("  (no result)", e)
P
}
i=0
for v in "MAS_FUSED_GN_STATS=0" "MAS_FUSED_GN_STATS=1" "MAS_FUSED_GN_STATS=0" "MAS_FUSED_GN_STATS=1"; do
  i=$((i+1)); n=$O/bench_$i.json
  echo "== bench $i [$v]"
  env $v $B > $n 2> $n.err
  short $n
done
