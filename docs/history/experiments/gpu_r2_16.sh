#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_16; mkdir -p $O
cd $R
KB="timeout 120 python tools/kbench.py"
{
for act in 0 2; do
  echo -n "wgrad tr  act=$act: "; MAS_WGRAD_DMA=0 $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1
  echo -n "wgrad dma act=$act: "; $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1
done
for s in "256 64" "512 32" "128 128" "512 16"; do set -- $s
  echo -n "c$1 hw$2 tr : "; MAS_WGRAD_DMA=0 $KB wgrad --n 32 --c $1 --hw $2 --act 2 | tail -1
  echo -n "c$1 hw$2 dma: "; $KB wgrad --n 32 --c $1 --hw $2 --act 2 | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_r2.py tests/test_gpu_spatial_attn.py -m gpu -q -x --timeout 600 2>&1 | tail -4
echo "== full"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | cut -c1-330
