#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
for rep in 1 2; do
for s in "128 256" "128 128" "256 64" "512 32"; do set -- $s
  for act in 2 0; do
    echo -n "new: "; $KB wgrad --n 32 --c $1 --hw $2 --act $act | tail -1
    echo -n "old: "; MAS_HIP_LIB=$V/wg_old.so $KB wgrad --n 32 --c $1 --hw $2 --act $act | tail -1
  done
done
done
echo "== bench old"; MAS_HIP_LIB=$V/wg_old.so timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
echo "== bench new"; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
