#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
KB="timeout 120 python tools/kbench.py"
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_model.py tests/test_gpu_parity_r2.py -m gpu -q --timeout 600 2>&1 | tail -3
{ echo -n "wide plain: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1; echo -n "wide act2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 2>&1 | tail -1; echo -n "wide act2 res: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 2>&1 | tail -1; echo -n "c512 hw32 act2: "; MAS_CONV_WIDE_MIN_TILES_PER_CU=1 $KB conv_fwd --n 32 --c 512 --hw 32 --act 2 2>&1 | tail -1; } | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['populations'], j['encoder_stack']['fwd_ms'])"
