#!/bin/bash
# round-3 GPU call 2: activation side output (mas_conv_fwd_act), reverse-order GroupNorm reductions, persistent wgrad scratch:
# correctness (full GPU suite), A/B benches of each switch, kbench of the touched kernels, kernel trace of the default step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_1; mkdir -p $O
cd $R
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -4 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
grep -h "encoder backward under\|fp32 copies=1 encoder.model.0\|act_out\|activation side output\|wgrad (LDS" $O/pytest_full.txt | head -40
B="timeout 300 python bench.py --no-cpu-baseline --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: plain %.4f ms gn_silu %.4f ms  frac %.4f  encoder %.3f ms" % (d["value"], d["ms_per_step"], r["plain"]["avg_launch_ms"], r["gn_silu"]["avg_launch_ms"], d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"]))
except Exception as e: print("  (no result)", e)
P
}
for v in default "MAS_CONV_ACT_OUT=0" "MAS_GN_REVERSE=0" "MAS_WGRAD_SCRATCH=0" "MAS_CONV_ACT_OUT=0 MAS_GN_REVERSE=0 MAS_WGRAD_SCRATCH=0" default; do
  tag=$(echo "$v" | tr ' =' '__'); n=$O/bench_$tag.json; [ -e $n ] && n=$O/bench_${tag}_2.json
  echo "== bench [$v]"
  if [ "$v" = default ]; then $B > $n 2> $n.err; else env $v $B > $n 2> $n.err; fi
  short $n
done
echo "== kbench"
KB="timeout 120 python tools/kbench.py"
{
for act in 0 2; do echo -n "wgrad act=$act: "; $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
echo -n "fwd act=2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 2>&1 | tail -1
echo -n "fwd act=0: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1
for r in 0 1; do
  echo -n "gn_bwd reverse=$r: "; MAS_GN_REVERSE=$r $KB gn_bwd --n 32 --c 128 --hw 256 2>&1 | tail -1
  echo -n "gn_stats reverse=$r: "; MAS_GN_REVERSE=$r $KB gn_stats --n 32 --c 128 --hw 256 2>&1 | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
echo "== trace"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -45 $O/kernel_trace_vq.txt
