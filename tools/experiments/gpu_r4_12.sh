#!/bin/bash
# round 4, session 12: spatial attention rewrite -- parity (wider shape list), model-level tests, step time, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spatial_attn.py tests/test_gpu_gn_coop.py tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -m gpu -x -q -rP > $O/pytest_sp.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_sp.txt; grep "rel-L2\|passed\|failed\|rc=" $O/pytest_sp.txt | grep -i "fwd rel\|passed\|failed\|rc=" | cut -c1-200
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz', r.get('package_power_w'), 'W; enc', d['encoder_stack']['fwd_ms'])"
done > $O/bench.txt 2>&1; cat $O/bench.txt
R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null
grep -i "spatial_attn\|total GPU" $O/kernel_trace_vq.txt | cut -c1-200
