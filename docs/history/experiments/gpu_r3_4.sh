#!/bin/bash
# round-3 GPU call 6: same-box A/B of the leaner wgrad kernel (variants/wgrad_r2.so = this tree with round 2's conv_wgrad_dma.hip)
# x activation side output on / off; kernel trace of the best configuration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_4; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
B="timeout 300 python bench.py --no-cpu-baseline --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: plain %.4f ms gn_silu %.4f ms  frac %.4f  encoder %.3f ms" % (d["value"], d["ms_per_step"], r["plain"]["avg_launch_ms"], r["gn_silu"]["avg_launch_ms"], d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"]))
except Exception as e: print("  (no result)", e)
P
}
i=0
for v in "MAS_CONV_ACT_OUT=0" "MAS_CONV_ACT_OUT=0 MAS_HIP_LIB=$V/wgrad_r2.so" "MAS_CONV_ACT_OUT=1" "MAS_CONV_ACT_OUT=1 MAS_HIP_LIB=$V/wgrad_r2.so" "MAS_CONV_ACT_OUT=0" "MAS_CONV_ACT_OUT=0 MAS_HIP_LIB=$V/wgrad_r2.so"; do
  i=$((i+1)); n=$O/bench_$i.json
  echo "== bench $i [$(echo $v | sed "s#$V/##")]"
  env $v $B > $n 2> $n.err
  short $n
done
echo "== trace (MAS_CONV_ACT_OUT=0)"
cd /tmp && export TMPDIR=/tmp
MAS_CONV_ACT_OUT=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -32 $O/kernel_trace_vq.txt
