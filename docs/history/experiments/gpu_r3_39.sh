#!/bin/bash
# round-3 GPU call 50: Downsample data gradient on conv_s2_dgrad_kernel vs zero-stuffing + the wide kernel (MAS_CONV_S2=0 switches all three s2 kernels)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -4
B="timeout 300 python bench.py --no-cpu-baseline --no-also --steps 15 --warmup 10"
for v in 1 0 1 0; do
  echo -n "bench [s2=$v]: "; MAS_CONV_S2=$v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f img/s  %.3f ms/step  dominant %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) | grep -E "s2_|zero_stuff|Li3ELi2E" | cut -c1-140
