#!/bin/bash
# round-3 GPU call 29: kernel trace of the VQ-IMG step after the split-K / GroupNorm changes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
tail -1 /tmp/pf_vq.log | cut -c1-200
python $R/tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $R/gpurun_out/r3_25_trace.txt > /dev/null
head -50 $R/gpurun_out/r3_25_trace.txt | cut -c1-170
