#!/bin/bash
# round 6, call 53: the GroupNorm backward's first pass out of the data-gradient convolution's epilogue (MAS_CONV_DGRAD_GNSTATS): parity, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_53; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_model.py -q -m gpu --timeout 600 2>&1 | tail -15 > $O/pytest.txt; tail -6 $O/pytest.txt
for rep in 1 2 3; do
  for on in 0 1; do
    MAS_CONV_DGRAD_GNSTATS=$on timeout 300 python bench.py --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('MAS_CONV_DGRAD_GNSTATS=$on', d['ms_per_step'], d['value'], d.get('final_loss'), {k:v['avg_launch_ms'] for k,v in r['populations'].items()}, r['frac'])"
  done
done | tee $O/ab.txt
