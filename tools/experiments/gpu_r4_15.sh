#!/bin/bash
# round 4, session 15: causal attention backward with the mask hoisted to one wave-uniform branch: parity, kbench, transformer step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transformer.py tests/test_gpu_parity_r2.py -m gpu -x -q -k "attn or attention or transformer or causal" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt | cut -c1-200
{ for i in 1 2; do python tools/kbench.py attn --n 8 --iters 50 2>&1 | grep "^attn"; done
  python tools/kbench.py attn --n 32 --iters 20 2>&1 | grep "^attn"; } > $O/kbench_attn.txt; cat $O/kbench_attn.txt
for i in 1 2; do timeout 300 python bench.py --workload transformer 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step; attn fwd frac', d['roofline']['frac'])"; done > $O/bench_tr.txt; cat $O/bench_tr.txt
