#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R; mkdir -p gpurun_out/r2_gaps
python tools/step_gaps.py $(find /tmp/pf_vq -name "*.db" | head -1) | tee gpurun_out/r2_gaps/step_gaps_vq.txt
