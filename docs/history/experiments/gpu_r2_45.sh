#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_transformer.py -m gpu -q --timeout 600 -k "colsum or linear" 2>&1 | tail -3
echo "== new Linear node"; timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
echo "== bf16 dW + cast"; MAS_LINEAR_FP32_DW=0 timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr -o tr -- python $R/bench.py --workload transformer --steps 3 --warmup 1 > /tmp/pf_tr.log 2>&1
cd $R; mkdir -p gpurun_out/r2_lin
python tools/rocprof_summary.py $(find /tmp/pf_tr -name "*.db" | head -1) gpurun_out/r2_lin/kernel_trace_transformer_linear.txt > /dev/null; head -30 gpurun_out/r2_lin/kernel_trace_transformer_linear.txt | cut -c1-150
