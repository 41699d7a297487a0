#!/bin/bash
# round 6, call 16: stream kernel XCD bands -- parity tests, step traffic, step time A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_16; mkdir -p $O
timeout 1200 python -m pytest -m gpu -q --timeout 600 tests/test_gpu_stream.py tests/test_gpu_stream_th8.py tests/test_gpu_parity_r4.py tests/test_gpu_model.py > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3; do for b in 0 1; do
  MAS_CONV_XCD_BANDS=$b timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('MAS_CONV_XCD_BANDS=$b', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done | tee $O/step_ab.txt
bash tools/step_traffic.sh $O/traffic 3 2 2>&1 | grep -E "TOTAL per step, every|^conv 3x3 s1 stream|dominant launch"
python3 tools/step_traffic.py $O/traffic 3 2 --json $O/step_traffic.json > $O/step_traffic.txt
