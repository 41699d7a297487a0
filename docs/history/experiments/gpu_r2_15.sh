#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_15; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spatial_attn.py -m gpu -q -rP --timeout 600 > $O/pytest_sp.txt 2>&1; grep -h "passed\|failed\|rel-L2\|^E " $O/pytest_sp.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -45 $O/kernel_trace_vq.txt | cut -c1-180
