#!/bin/bash
# round-3 GPU call 16: full validation of the shipped state: GPU suite, smoke, default bench line, kernel trace, PMC of the dominant kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_13; mkdir -p $O
cd $R
echo "== pytest"
timeout 1800 python -m pytest tests -m gpu -q -rA > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
grep -h "^FAILED\|^ERROR" $O/pytest_full.txt | head -20
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/smoke.txt
echo "== default bench"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-700 $O/bench_default.json; echo
echo "== trace"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -30 $O/kernel_trace_vq.txt
python tools/step_gaps.py $(find /tmp/pf_vq -name "*.db" | head -1) > $O/step_gaps.txt 2>&1; head -3 $O/step_gaps.txt
echo "== PMC dominant kernel (plain wide, 128->128 @256^2 x 32)"
export GRAFT_REPO_ROOT=$R
bash tools/pmc_kernel.sh FETCH_SIZE conv_fwd --n 32 --c 128 --hw 256 2>&1 | grep -v amdgpu.ids | tee $O/pmc_fetch.txt
bash tools/pmc_kernel.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" conv_fwd --n 32 --c 128 --hw 256 2>&1 | grep -v amdgpu.ids | tee $O/pmc_write.txt
echo "== PMC wgrad (LDS conflicts, MFMA busy)"
bash tools/pmc_kernel.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" wgrad --n 32 --c 128 --hw 256 2>&1 | grep -v amdgpu.ids | tee $O/pmc_wgrad.txt
