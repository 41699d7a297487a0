#!/bin/bash
# round 5, call 9: (a) the transformer's folded small launches (one-stage LayerNorm parameter fold, Linear bias gradient out of the LayerNorm
# backward): tests, the transformer bench line, its kernel trace; (b) the block budget of the element-wise GroupNorm passes against torch.add
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_transformer.py tests/test_abi.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --workload transformer --steps 10 --warmup 4 2>/dev/null | grep '^{' > $O/bench_transformer.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_9/bench_transformer.json").read().strip().splitlines()[-1])
print("transformer:", d["value"], d["unit"], d["ms_per_step"], "ms/step")
PY
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr -o tr -- python $R/bench.py --workload transformer --steps 3 --warmup 1 > /tmp/pf_tr.log 2>&1
cd $R && python tools/rocprof_summary.py $(find /tmp/pf_tr -name "*.db" | head -1) $O/kernel_trace_transformer.txt > /dev/null 2>&1; head -34 $O/kernel_trace_transformer.txt | cut -c1-150
timeout 600 python tools/probes/gn_blocks_probe.py 2>&1 | grep -v "Warning\|warn" | tee $O/gn_blocks.txt
