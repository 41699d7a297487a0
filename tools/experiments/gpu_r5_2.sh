#!/bin/bash
# round 5, call 2: Upsample + conv in its sub-pixel form (conv_up2.hip): parity (pack image, small / multi-tile / benched shapes), kbench of the
# three Upsample forwards, the bench line, the step's kernel trace; GroupNorm tests after the queue kernel was shelved
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_up2.py -x -q -s > $O/pytest_up2.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_up2.txt
grep -v "^  warn\|Warning\|amdgpu.ids" $O/pytest_up2.txt | tail -60 | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_gn_coop.py tests/test_abi.py -x -q > $O/pytest_gn.txt 2>&1; tail -3 $O/pytest_gn.txt
for v in 1 0; do
  export MAS_CONV_UP2=$v
  for cfg in "128 128" "256 64" "512 32"; do set -- $cfg
    timeout 120 python tools/kbench.py conv_fwd --c $1 --hw $2 --ups 1 --stats 1 --iters 200 2>&1 | tail -1 | sed "s/^/MAS_CONV_UP2=$v /"
  done
done > $O/kbench_up2.txt 2>&1; cat $O/kbench_up2.txt
unset MAS_CONV_UP2
timeout 600 python bench.py --no-cpu-baseline --no-also > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-400
MAS_CONV_UP2=0 timeout 600 python bench.py --no-cpu-baseline --no-also > $O/bench_off.txt 2>&1; tail -1 $O/bench_off.txt | cut -c1-400
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -60 $O/kernel_trace_vq.txt | cut -c1-200
