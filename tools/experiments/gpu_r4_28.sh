#!/bin/bash
# round 4, session 28: spatial attention with 8 waves per work-group (-DSP_NW=8 variant build) against 4: parity, kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_28; mkdir -p $O; R=$PWD
for v in default spnw8; do
  [ $v = default ] && unset MAS_HIP_LIB || export MAS_HIP_LIB=$R/make-a-scene_amd/csrc/build/variants/$v.so
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_spatial_attn.py -x -q 2>&1 | tail -1
  for hw in 16 8; do
    cd /tmp; rm -rf /tmp/pf_sp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_sp -o sp -- python $R/tools/kbench.py sp_attn --hw $hw --iters 30 > /tmp/pf_sp.log 2>&1; cd $R
    python tools/rocprof_summary.py $(find /tmp/pf_sp -name "*.db" | head -1) /tmp/trace_sp.txt > /dev/null; grep -i "spatial" /tmp/trace_sp.txt | cut -c1-120 | sed "s/^/hw=$hw /"
  done
done > $O/ab.txt 2>&1; cat $O/ab.txt
