#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
echo "== replay"; timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /tmp/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['library_gemm_selection'])"; tail -2 /tmp/err.txt
echo "== off"; MAS_BENCH_TUNABLEOP=0 timeout 600 python bench.py --workload transformer --steps 10 --warmup 5 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['library_gemm_selection'])"
echo "== e2e"; timeout 600 python bench.py --workload e2e --steps 4 --warmup 2 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['library_gemm_selection'])"
git status --short 2>/dev/null | head -3; ls make-a-scene_amd/tuning
