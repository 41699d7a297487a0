#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/d; mkdir -p $O
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
for cfg in "new:" "newcore:$V/s_core.so" "old:STREAM0" "oldcore:$V/oldcore.so:STREAM0"; do
  name=${cfg%%:*}; rest=${cfg#*:}
  unset MAS_HIP_LIB MAS_CONV_STREAM
  case "$rest" in *STREAM0*) export MAS_CONV_STREAM=0;; esac
  lib=${rest%%:*}; if [ -n "$lib" ] && [ "$lib" != "STREAM0" ]; then export MAS_HIP_LIB=$lib; fi
  echo "== $name"
  bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" conv_fwd --n 32 --c 128 --hw 256 2>&1 | grep -v amdgpu.ids | tail -3
done 2>&1 | tee $O/clock.txt
