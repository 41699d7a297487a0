#!/bin/bash
# Tunes the library GEMM selection (PyTorch TunableOp: hipBLASLt / rocBLAS solutions per shape) for bench.py --workload transformer on
# this box and writes gpurun_out/r2_tunable/tunableop_results0.csv -- copy it to make-a-scene_amd/tuning/tunableop_gfx950_transformer.csv.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r2_tunable; mkdir -p $O; rm -f $O/*.csv
J='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'
export MAS_BENCH_TUNABLEOP=0                      # bench.py leaves TunableOp alone: the environment below drives it
echo "== library default selection"; timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "$J"
echo "== tuning run"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunableop_results.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 PYTORCH_TUNABLEOP_VERBOSE=0
( time timeout 1500 python bench.py --workload transformer --steps 10 --warmup 5 2> $O/tune.err | python -c "$J" ) 2>&1 | tail -5
wc -l $O/*.csv
echo "== replay (no tuning)"
export PYTORCH_TUNABLEOP_TUNING=0
timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "$J"
