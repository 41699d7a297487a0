#!/bin/bash
# round-3 GPU call 5: leaner conv_wgrad_dma (lane-constant DMA offsets, base + immediate fragment reads): correctness, kbench, bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_3; mkdir -p $O
cd $R
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
echo "== kbench"
KB="timeout 120 python tools/kbench.py"
{
for act in 0 2; do echo -n "wgrad act=$act: "; $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
for s in "128 128" "256 64" "512 32" "512 16"; do set -- $s
  for act in 0 2; do echo -n "wgrad c$1 hw$2 act=$act: "; $KB wgrad --n 32 --c $1 --hw $2 --act $act 2>&1 | tail -1; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: plain %.4f ms gn_silu %.4f ms  frac %.4f  encoder %.3f ms" % (d["value"], d["ms_per_step"], r["plain"]["avg_launch_ms"], r["gn_silu"]["avg_launch_ms"], d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"]))
except Exception as e: print("  (no result)", e)
P
}
for v in "MAS_CONV_ACT_OUT=0" "MAS_CONV_ACT_OUT=1" "MAS_CONV_ACT_OUT=0"; do
  tag=$(echo "$v" | tr ' =' '__'); n=$O/bench_$tag.json; [ -e $n ] && n=$O/bench_${tag}_2.json
  echo "== bench [$v]"
  env $v $B > $n 2> $n.err
  short $n
done
