#!/bin/bash
# round 4, session 18: HBM traffic of the dominant kernel as it runs in the training forward (statistics epilogue on) and as the
# data gradient (plain), rocprofv3 --pmc in separate passes (tools/pmc_kernel.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_18; mkdir -p $O
{ for st in 1 0; do
  echo "== conv_fwd 128->128 @256^2 B=32, stats=$st"
  bash tools/pmc_kernel.sh "FETCH_SIZE" conv_fwd --n 32 --c 128 --hw 256 --stats $st
  bash tools/pmc_kernel.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" conv_fwd --n 32 --c 128 --hw 256 --stats $st
done
echo "== wgrad 128->128 @256^2 B=32"
bash tools/pmc_kernel.sh "FETCH_SIZE" wgrad --n 32 --c 128 --hw 256
bash tools/pmc_kernel.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" wgrad --n 32 --c 128 --hw 256
} 2>&1 | grep -v "^$" | cut -c1-330 > $O/pmc_traffic.txt; cat $O/pmc_traffic.txt
