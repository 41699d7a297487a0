#!/bin/bash
# round 5, call 4: new parity tests (sub-pixel Upsample at the benched shapes, ResnetBlock with shortcut, distinct-image B = 16 backward vs the oracle,
# Adam edge cases); gn_act walking the images last to first (MAS_GN_ACT_REV) with / without non-temporal stores: same-box step A/B; the N>1 code path on one rank
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_up2.py tests/test_gpu_parity_r5.py tests/test_gpu_adam.py -q -s > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt
grep -v "^  warn\|Warning\|amdgpu.ids" $O/pytest_new.txt | grep "rel-L2\|passed\|failed\|FAIL\|Error\|assert\|rc=\|dtype" | tail -60 | cut -c1-330
{
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 0"; do set -- $cfg
  echo "== MAS_GN_ACT_REV=$1 MAS_GN_ACT_NT=$2 (rep $rep)"; MAS_GN_ACT_REV=$1 MAS_GN_ACT_NT=$2 timeout 300 python bench.py --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; encoder fwd', d['encoder_stack']['fwd_ms'], 'ms; dominant launch', d['roofline']['avg_launch_ms'], 'clock', d['roofline'].get('sustained_clock_mhz'))"
done; done
} > $O/gn_act_rev.txt 2>&1; cat $O/gn_act_rev.txt
MAS_BENCH_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack --dp ddp --steps 5 --warmup 3 > $O/ddp_one_rank.txt 2>&1; tail -5 $O/ddp_one_rank.txt | cut -c1-300
