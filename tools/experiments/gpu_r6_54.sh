#!/bin/bash
# round 6, call 54: last check of the final tree: full GPU suite, smoke, one bench line, a short soak
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6_54; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -1 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 600 python bench.py 2>/dev/null | grep '^{' > $O/bench.json; python3 -c "import json; d=json.loads(open('$O/bench.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('final_loss'), {k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('also',{}).items()})"
MAS_WGRAD_STREAM=1 CHECK=0 TRIALS=100 timeout 900 python tools/experiments/side_stream_stress2.py 2>&1 | grep -v "Warn\|amdgpu.ids\|detach\|return float" | tee $O/soak.txt
