#!/bin/bash
# round-2 GPU call 10: new sampling / token / bench-N>1 tests, full suite, transformer + e2e workloads, PMC traffic of the wide kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_10; mkdir -p $O
cd $R
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rP --timeout 900 > $O/pytest_full.txt 2>&1; tail -4 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR\|cached decode vs" $O/pytest_full.txt | head -20
echo "== bench transformer"
timeout 600 python bench.py --workload transformer --steps 5 --warmup 2 > $O/bench_transformer.json 2> $O/bench_transformer.err; cut -c1-1500 $O/bench_transformer.json; tail -2 $O/bench_transformer.err
echo "== bench e2e"
timeout 900 python bench.py --workload e2e --steps 2 --warmup 1 > $O/bench_e2e.json 2> $O/bench_e2e.err; cut -c1-1500 $O/bench_e2e.json; tail -2 $O/bench_e2e.err
echo "== bench vq (with cpu baseline)"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-2600 $O/bench.json; tail -2 $O/bench.err
echo "== pmc traffic"
for cnt in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  for act in 0 2; do echo "== act=$act $cnt"; bash tools/pmc_kernel.sh "$cnt" conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | grep -v amdgpu.ids | grep -i "wide\|error" | tail -2; done
done 2>&1 | tee $O/pmc_traffic.txt
