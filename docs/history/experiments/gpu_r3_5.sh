#!/bin/bash
# round-3 GPU call 7: packed SiLU, 64-cout tiles for the small maps, unrolled GroupNorm finalizes, fragment prefetch in the prologue
# variant of the wgrad kernel: correctness + same-box A/B (variants: wgrad_nopf.so = no prefetch in the prologue variant)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_5; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
echo "== kbench"
KB="timeout 120 python tools/kbench.py"
{
echo -n "wgrad act=2: "; $KB wgrad --n 32 --c 128 --hw 256 --act 2 | tail -1
echo -n "wgrad act=2 (no prefetch): "; MAS_HIP_LIB=$V/wgrad_nopf.so $KB wgrad --n 32 --c 128 --hw 256 --act 2 | tail -1
echo -n "fwd act=2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 2>&1 | tail -1
for bc in 1 0; do
  echo -n "conv c512 hw16 BC64=$bc: "; MAS_CONV_BC64=$bc $KB conv_fwd --n 32 --c 512 --hw 16 2>&1 | tail -1
  echo -n "conv c512 hw16 act=2 BC64=$bc: "; MAS_CONV_BC64=$bc $KB conv_fwd --n 32 --c 512 --hw 16 --act 2 2>&1 | tail -1
  echo -n "1x1 c512 hw16 BC64=$bc: "; MAS_CONV_BC64=$bc $KB conv_fwd --n 32 --c 512 --hw 16 --ks 1 2>&1 | tail -1
  echo -n "stride2 c512 hw32 BC64=$bc: "; MAS_CONV_BC64=$bc $KB conv_fwd --n 32 --c 512 --hw 32 --stride 2 2>&1 | tail -1
done
echo -n "gn_bwd c512 hw16: "; $KB gn_bwd --n 32 --c 512 --hw 16 --iters 30 2>&1 | tail -1
echo -n "gn_stats c512 hw16: "; $KB gn_stats --n 32 --c 512 --hw 16 --iters 30 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: plain %.4f ms gn_silu %.4f ms  frac %.4f  encoder %.3f ms" % (d["value"], d["ms_per_step"], r["plain"]["avg_launch_ms"], r["gn_silu"]["avg_launch_ms"], d["roofline"]["frac"], d["encoder_stack"]["fwd_ms"]))
except Exception as e: print("  (no result)", e)
P
}
i=0
for v in "X=1" "MAS_HIP_LIB=$V/wgrad_nopf.so" "MAS_CONV_BC64=0" "MAS_HIP_LIB=$V/wgrad_r2.so" "X=1"; do
  i=$((i+1)); n=$O/bench_$i.json
  echo "== bench $i [$(echo $v | sed "s#$V/##")]"
  env $v $B > $n 2> $n.err
  short $n
done
echo "== full default bench line (with cpu baseline and the also block)"
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; python - $O/bench_full.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("host_physical_cores")); print(json.dumps(d.get("also"))[:1500])
P
