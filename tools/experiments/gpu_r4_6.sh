#!/bin/bash
# round 4, session 6: tiled pack parity, round-4 parity incl. N = 160, clock / power on the right card, step kernel trace, attention PMC after the LPT fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pack_tiles.py tests/test_gpu_parity_r4.py -q -s > $O/pytest_a.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_a.txt
grep -v "^  warn\|Warning\|amdgpu.ids\|^  bf16\|^  fp32" $O/pytest_a.txt | tail -40 | cut -c1-220
bash tools/probes/build.sh > /dev/null 2>&1
timeout 600 python tools/probes/clock_power.py > $O/clock_power.txt 2>&1; cat $O/clock_power.txt | cut -c1-220
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -45 $O/kernel_trace_vq.txt | cut -c1-200
{
echo "# round 4: rocprofv3 --pmc (three separate passes, tools/pmc_kernel.sh) on tools/kbench.py attn --n 8 (B=8, H=16, S=1536, hd=64, bf16), heaviest-first dispatch (MAS_ATTN_LPT=1, the default since round 3)"
bash tools/pmc_kernel.sh "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" attn --n 8
bash tools/pmc_kernel.sh "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" attn --n 8
bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" attn --n 8
} > $O/attn_pmc.txt 2>&1; cat $O/attn_pmc.txt | cut -c1-260
