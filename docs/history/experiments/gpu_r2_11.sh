#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_11; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_sampling.py -m gpu -q 2>&1 | tail -2
echo "== bench e2e"
timeout 900 python bench.py --workload e2e --steps 2 --warmup 1 > $O/bench_e2e.json 2> $O/bench_e2e.err; cut -c1-1500 $O/bench_e2e.json; tail -2 $O/bench_e2e.err
echo "== fetch calibration"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc; timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fc -o fc --output-format csv -- $R/tools/probes/fetch_calib > $O/fetch_calib.txt 2>&1
python3 - >> $O/fetch_calib.txt <<'PY'
import csv, glob
f = glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"], "KB")
PY
cat $O/fetch_calib.txt | grep -v amdgpu.ids | tail -8
