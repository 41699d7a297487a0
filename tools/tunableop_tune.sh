#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r2_tunable; mkdir -p $O
echo "== baseline"; timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
echo "== tunableop tuning run"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunableop_results.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 PYTORCH_TUNABLEOP_VERBOSE=0
( time timeout 1500 python bench.py --workload transformer --steps 10 --warmup 5 2> $O/tune.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])" ) 2>&1 | tail -5
ls -la $O; wc -l $O/*.csv
echo "== tunableop replay (no tuning)"
export PYTORCH_TUNABLEOP_TUNING=0
timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
