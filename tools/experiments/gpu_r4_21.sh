#!/bin/bash
# round 4, session 21: one-launch Adam (optim.hip / mas_hip.optim.Adam): parity vs torch.optim.Adam, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_adam.py tests/test_gpu_parity_r2.py -x -q -k "adam or optimizer or state_dict or unsupported" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -8 $O/pytest.txt | cut -c1-220
for o in mas torch mas torch; do
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also --optimizer $o 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('optimizer=$o', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('avg_launch_ms'), 'clock', r.get('sustained_clock_mhz'), 'MHz; loss', d['final_loss'])"
done > $O/bench.txt 2>&1; cat $O/bench.txt
for o in mas torch; do timeout 300 python bench.py --workload transformer --optimizer $o 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('optimizer=$o', d['value'], d['unit'], d['ms_per_step'], 'ms/step')"; done > $O/bench_tr.txt; cat $O/bench_tr.txt
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
