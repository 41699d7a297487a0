#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
echo "== fp32-out dW"; timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
echo "== bf16 dW + cast"; MAS_LINEAR_FP32_DW=0 timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
echo "== bf16 dW + cast, no tunableop"; MAS_BENCH_TUNABLEOP=0 MAS_LINEAR_FP32_DW=0 timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
MAS_LINEAR_FP32_DW=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr -o tr -- python $R/bench.py --workload transformer --steps 3 --warmup 1 > /tmp/pf_tr.log 2>&1
cd $R; mkdir -p gpurun_out/r2_lin
python tools/rocprof_summary.py $(find /tmp/pf_tr -name "*.db" | head -1) gpurun_out/r2_lin/kernel_trace_transformer_linear.txt > /dev/null; head -24 gpurun_out/r2_lin/kernel_trace_transformer_linear.txt | cut -c1-200
