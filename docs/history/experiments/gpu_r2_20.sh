#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py -m gpu -q -x --timeout 600 -k fused 2>&1 | grep -E "^E |assert|Error" | head -12
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R
mkdir -p gpurun_out/r2_20
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) gpurun_out/r2_20/kernel_trace_vq.txt > /dev/null 2>&1; head -24 gpurun_out/r2_20/kernel_trace_vq.txt | cut -c1-150
