#!/bin/bash
# round-3 GPU call 10: fused epilogue statistics now that the producers of GroupNorm inputs are the PLAIN wide variants (no spills)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_8; mkdir -p $O
cd $R
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
for v in "MAS_FUSED_GN_STATS=0" "MAS_FUSED_GN_STATS=1" "MAS_FUSED_GN_STATS=0" "MAS_FUSED_GN_STATS=1"; do
  i=$((i+1)); n=$O/bench_$i.json
  echo "== bench $i [$v]"
  env $v $B > $n 2> $n.err
  short $n
done
echo "== parity with fused statistics ON"
MAS_FUSED_GN_STATS=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -m gpu -q 2>&1 | tail -3
echo "== trace (default)"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -34 $O/kernel_trace_vq.txt
