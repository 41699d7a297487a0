#!/bin/bash
# round-3 GPU call 12: MAS_OVERLAP=1 (GroupNorm backward on G CUs || weight gradient on 256 - G, ResnetBlock backward): parity + A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_9; mkdir -p $O
cd $R
echo "== parity with MAS_OVERLAP=1"
MAS_OVERLAP=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_dp.py -m gpu -q 2>&1 | tail -4
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
short() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]["populations"]
    print("  %.2f img/s  %.3f ms/step  dominant: %s  frac %.4f  loss %.5f" % (d["value"], d["ms_per_step"], {k:(v["launches"], v["avg_launch_ms"]) for k,v in r.items()}, d["roofline"]["frac"], d["final_loss"]))
except Exception as e: print("  (no result)", e)
P
}
i=0
for v in "MAS_OVERLAP=0" "MAS_OVERLAP=1 MAS_OVERLAP_GN_CUS=96" "MAS_OVERLAP=1 MAS_OVERLAP_GN_CUS=112" "MAS_OVERLAP=1 MAS_OVERLAP_GN_CUS=128" "MAS_OVERLAP=1 MAS_OVERLAP_GN_CUS=144" "MAS_OVERLAP=0"; do
  i=$((i+1)); n=$O/bench_$i.json
  echo "== bench $i [$v]"
  env $v $B > $n 2> $n.err
  short $n; tail -2 $n.err | grep -v amdgpu.ids
done
