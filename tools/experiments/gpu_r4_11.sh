#!/bin/bash
# round 4, session 11: per-kernel times of the spatial attention kernels (kernel trace of tools/kbench.py sp_attn) + phase stamps
cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=make-a-scene_amd TMPDIR=/tmp
O=gpurun_out/r4_11; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_gpu_spatial_attn.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-200 > $O/pytest.txt; cat $O/pytest.txt
for hw in 16 8; do
cd /tmp; rm -rf /tmp/pf_sp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_sp -o sp -- python $R/tools/kbench.py sp_attn --hw $hw --iters 30 > /tmp/pf_sp.log 2>&1; cd $R
python tools/rocprof_summary.py $(find /tmp/pf_sp -name "*.db" | head -1) $O/trace_sp_$hw.txt > /dev/null; grep -i "spatial\|total" $O/trace_sp_$hw.txt | cut -c1-160
done
MAS_HIP_LIB=make-a-scene_amd/csrc/build/variants/sptrace.so python tools/kbench.py sp_attn --iters 50 2>&1 | grep -A3 "sp_attn forward phases" | tee $O/phases.txt
