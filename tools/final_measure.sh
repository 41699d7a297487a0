#!/bin/bash
# End-of-round evidence run on one MI355X box: tests, smoke, bench, single-kernel numbers, rocprofv3 kernel traces.
# Writes everything under gpurun_out/final/ (copy the summaries to profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
{
  for act in 0 2; do timeout 100 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act $act | tail -1; done
  timeout 100 python tools/kbench.py dgrad --n 32 --c 128 --hw 256 | tail -1
  for act in 0 2; do timeout 100 python tools/kbench.py wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
  timeout 100 python tools/kbench.py gn_stats --n 32 --c 128 --hw 256 | tail -1
  timeout 100 python tools/kbench.py gn_bwd --n 32 --c 128 --hw 256 | grep "^gn_bwd"
  timeout 100 python tools/kbench.py vq --n 32 | tail -1
  timeout 100 python tools/kbench.py conv_fwd --n 32 --c 512 --hw 32 | tail -1
  timeout 100 python tools/kbench.py wgrad --n 32 --c 512 --hw 32 | tail -1
  timeout 100 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 64 | tail -1
  timeout 100 python tools/kbench.py wgrad --n 32 --c 256 --hw 64 | tail -1
  timeout 100 python tools/kbench.py attn --n 8 | tail -2
  timeout 100 python tools/kbench.py attn --n 32 | tail -2
  timeout 200 python tools/bench_transformer.py --batch 8 | tail -1
  timeout 60 tools/probes/mfma_peak
} > $O/kbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr -o tr -- python $R/tools/bench_transformer.py --batch 8 --steps 3 > /tmp/pf_tr.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null
python tools/rocprof_summary.py $(find /tmp/pf_tr -name "*.db" | head -1) $O/kernel_trace_transformer.txt > /dev/null
tail -2 $O/pytest_gpu.txt; cat $O/smoke.txt | tail -1; cut -c1-300 $O/bench.json
