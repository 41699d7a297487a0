#!/bin/bash
# round 6, call 27: MAS_WGRAD_STREAM = 0 / 1 / 2 step A/B + timeline of mode 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_27; mkdir -p $O
MAS_WGRAD_STREAM=2 timeout 900 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_parity_r5.py -k "resnet or decoder" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2; do for b in 0 1 2; do
  MAS_WGRAD_STREAM=$b timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_WGRAD_STREAM=$b', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['final_loss'])"
done; done | tee $O/step_ab.txt
cd /tmp
MAS_WGRAD_STREAM=2 timeout 400 rocprofv3 --kernel-trace -d /tmp/ov -o ov -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/ov.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/overlap_timeline.py $(find /tmp/ov -name "*.db" | head -1) > $O/timeline2.txt 2>&1; head -22 $O/timeline2.txt
