#!/bin/bash
# round 6, call 61: the codebook reservoir's permutations without a host synchronisation (MAS_RESERVOIR_ASYNC): tests, step A/B, idle gaps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_61; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py -q -m gpu --timeout 600 2>&1 | tail -2 | tee $O/pytest.txt
for rep in 1 2 3; do
  for on in 0 1; do
    MAS_RESERVOIR_ASYNC=$on timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_RESERVOIR_ASYNC=$on', d['ms_per_step'], d['value'], d.get('final_loss'))"
  done
done | tee $O/ab.txt
cd /tmp
for on in 0 1; do
  rm -rf /tmp/gp$on; MAS_RESERVOIR_ASYNC=$on timeout 400 rocprofv3 --kernel-trace -d /tmp/gp$on -o gp -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/gp.log 2>&1
  echo "== MAS_RESERVOIR_ASYNC=$on"; python3 $GRAFT_REPO_ROOT/tools/step_gaps.py $(find /tmp/gp$on -name "*.db" | head -1) 12
done | tee $O/gaps.txt
