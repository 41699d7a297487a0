#!/bin/bash
# full GPU suite + bench with the wide kernel dispatched by default
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_9; mkdir -p $O
cd $R
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rP --timeout 900 > $O/pytest_full.txt 2>&1; tail -5 $O/pytest_full.txt
grep -h "img256 bf16 vs\|fwd plain\|fwd GN\|^dgrad:\|wgrad act\|^FAILED\|^ERROR" $O/pytest_full.txt | head -30
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-1700 $O/bench.json; tail -3 $O/bench.err
echo "== bench stream only"
MAS_CONV_WIDE=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err; cut -c1-330 $O/bench_stream.json
echo "== trace"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -24 $O/kernel_trace_vq.txt | cut -c1-200
