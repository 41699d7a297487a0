#!/bin/bash
# round-3 GPU call 4: side-output stores moved behind the weight DMA (counted wait 5); Infinity-Cache behaviour of the GroupNorm
# passes as a function of the tensor size (kbench sweep over the batch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_2; mkdir -p $O
cd $R
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
B="timeout 300 python bench.py --no-cpu-baseline --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: plain %.4f ms gn_silu %.4f ms  frac %.4f  encoder %.3f ms" % (d["value"], d["ms_per_step"], r["plain"]["avg_launch_ms"], r["gn_silu"]["avg_launch_ms"], d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"]))
except Exception as e: print("  (no result)", e)
P
}
for v in default "MAS_CONV_ACT_OUT=0" default "MAS_CONV_ACT_OUT=0"; do
  tag=$(echo "$v" | tr ' =' '__'); n=$O/bench_$tag.json; [ -e $n ] && n=$O/bench_${tag}_2.json
  echo "== bench [$v]"
  if [ "$v" = default ]; then $B > $n 2> $n.err; else env $v $B > $n 2> $n.err; fi
  short $n
done
echo "== kbench: GroupNorm passes vs tensor size (c=128 hw=256: 16.8 MB per image)"
KB="timeout 120 python tools/kbench.py"
{
for n in 1 2 4 8 16 32; do
  echo -n "n=$n: "; $KB gn_bwd --n $n --c 128 --hw 256 --iters 30 2>&1 | tail -1
  echo -n "n=$n: "; $KB gn_stats --n $n --c 128 --hw 256 --iters 30 2>&1 | tail -1
done
for s in "512 32" "512 16" "256 64" "128 128"; do set -- $s
  echo -n "c$1 hw$2: "; $KB gn_bwd --n 32 --c $1 --hw $2 --iters 30 2>&1 | tail -1
  echo -n "c$1 hw$2: "; $KB gn_stats --n 32 --c $1 --hw $2 --iters 30 2>&1 | tail -1
done
echo -n "fwd act=2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 2>&1 | tail -1
for s in "512 16" "512 32" "256 64"; do set -- $s
  echo -n "conv c$1 hw$2: "; $KB conv_fwd --n 32 --c $1 --hw $2 2>&1 | tail -1
  echo -n "dgrad c$1 hw$2: "; $KB dgrad --n 32 --c $1 --hw $2 2>&1 | tail -1
  echo -n "wgrad c$1 hw$2: "; $KB wgrad --n 32 --c $1 --hw $2 2>&1 | tail -1
done
echo -n "stride2 c128 hw256: "; $KB conv_fwd --n 32 --c 128 --hw 256 --stride 2 2>&1 | tail -1
echo -n "stride2 c512 hw32: "; $KB conv_fwd --n 32 --c 512 --hw 32 --stride 2 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
