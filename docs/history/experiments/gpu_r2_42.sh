#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 100 python tools/kbench.py"
MAS_HIP_LIB=$V/wide_stagger.so timeout 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_kernels.py -m gpu -q --timeout 600 2>&1 | tail -2
for rep in 1 2; do
for s in "128 256" "128 128" "256 64"; do set -- $s
  echo -n "base:    "; $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 | tail -1
  echo -n "stagger: "; MAS_HIP_LIB=$V/wide_stagger.so $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 | tail -1
done
echo -n "base res:    "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 | tail -1
echo -n "stagger res: "; MAS_HIP_LIB=$V/wide_stagger.so $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 | tail -1
done
echo "== bench base"; timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
echo "== bench stagger"; MAS_HIP_LIB=$V/wide_stagger.so timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
