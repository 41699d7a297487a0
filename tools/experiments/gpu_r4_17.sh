#!/bin/bash
# round 4, session 17: stride-2 kernels on the 32 -> 16 Downsample (half-wide tiles): parity, kbench, encoder stack, step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_model.py -x -q -k "stride2 or round3_kernels or model or encoder or decoder" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt | cut -c1-200
{ timeout 120 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 32 --stride 2 --iters 50 2>&1 | grep "^conv_fwd"
  timeout 120 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 64 --stride 2 --iters 50 2>&1 | grep "^conv_fwd"; } > $O/kbench.txt 2>&1; cat $O/kbench.txt
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz; enc', d['encoder_stack']['fwd_ms'])"
done > $O/bench.txt 2>&1; cat $O/bench.txt
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null
python tools/step_gaps.py $(find /tmp/pf_vq -name "*.db" | head -1) > $O/step_gaps.txt 2>&1; head -12 $O/step_gaps.txt
grep -i "s2\|conv_fwd_kernel\|total GPU" $O/kernel_trace_vq.txt | cut -c1-180
