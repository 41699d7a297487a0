#!/bin/bash
# End-of-round evidence run on one MI355X box: tests, smoke, bench lines, single-kernel numbers, rocprofv3 kernel traces.
# Writes everything under gpurun_out/final/ (copy the summaries to profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --workload transformer > $O/bench_transformer.json 2> $O/bench_transformer.err
timeout 600 python bench.py --workload e2e > $O/bench_e2e.json 2> $O/bench_e2e.err
[ -x tools/probes/mfma_peak ] || bash tools/probes/build.sh > /dev/null 2>&1
KB="timeout 100 python tools/kbench.py"
{
  for act in 0 2; do $KB conv_fwd --n 32 --c 128 --hw 256 --act $act | tail -1; done
  $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 | tail -1
  $KB dgrad --n 32 --c 128 --hw 256 | tail -1
  for act in 0 2; do $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
  echo "-- the same launches on round 1's kernels (MAS_CONV_WIDE=0 MAS_CONV_STREAM=0 MAS_WGRAD_DMA=0)"
  for act in 0 2; do MAS_CONV_WIDE=0 MAS_CONV_STREAM=0 $KB conv_fwd --n 32 --c 128 --hw 256 --act $act | tail -1; done
  for act in 0 2; do MAS_WGRAD_DMA=0 $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
  echo "-- other shapes"
  $KB gn_stats --n 32 --c 128 --hw 256 | tail -1
  $KB gn_bwd --n 32 --c 128 --hw 256 --three 1 | grep "^gn_bwd"
  $KB vq --n 32 | tail -1
  for s in "512 32" "256 64" "128 128" "256 128" "512 64"; do set -- $s
    $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 | tail -1; $KB wgrad --n 32 --c $1 --hw $2 --act 2 | tail -1
  done
  echo "-- Upsample + conv (sub-pixel form: conv_up2.hip, conv_wgrad_dma KS = 2), forward with statistics / weight gradient; then the same on the 3x3 kernels"
  for s in "128 128" "256 64" "512 32"; do set -- $s
    $KB conv_fwd --n 32 --c $1 --hw $2 --ups 1 --stats 1 | tail -1; $KB wgrad --n 32 --c $1 --hw $2 --ups 1 | tail -1
  done
  for s in "128 128" "256 64" "512 32"; do set -- $s
    MAS_CONV_UP2=0 $KB conv_fwd --n 32 --c $1 --hw $2 --ups 1 --stats 1 | tail -1; MAS_CONV_UP2=0 $KB wgrad --n 32 --c $1 --hw $2 --ups 1 | tail -1
  done
  $KB conv_fwd --n 32 --c 128 --hw 256 --stride 2 | tail -1
  $KB sp_attn --iters 50 | grep sp_attn
  $KB gn_bwd --n 32 --c 128 --hw 256 --res 1 --three 1 | grep "^gn_bwd"
  $KB gn_fwd --c 512 --hw 16 | grep gn_fwd
  $KB attn --n 8 | tail -2
  $KB attn --n 32 | tail -2
  timeout 60 tools/probes/mfma_peak
} 2>&1 | grep -v amdgpu.ids > $O/kbench.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/pf_vq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_tr -o tr -- python $R/bench.py --workload transformer --steps 3 --warmup 1 > /tmp/pf_tr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_enc -o enc -- python $R/tools/enc_fwd.py > /tmp/pf_enc.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null
python tools/rocprof_summary.py $(find /tmp/pf_tr -name "*.db" | head -1) $O/kernel_trace_transformer.txt > /dev/null
python tools/rocprof_summary.py $(find /tmp/pf_enc -name "*.db" | head -1) $O/kernel_trace_encoder_fwd.txt > /dev/null
tail -2 $O/pytest_gpu.txt; tail -1 $O/smoke.txt; cut -c1-300 $O/bench.json; cut -c1-300 $O/bench_transformer.json; head -12 $O/kbench.txt
# what whole steps move across the fabric (rocprofv3 --pmc, own passes): copy traffic/.. summary + json to profiles/ if the tree changed the kernels
bash tools/step_traffic.sh gpurun_out/final/traffic 3 2 > $O/step_traffic_run.txt 2>&1; python3 tools/step_traffic.py gpurun_out/final/traffic 3 2 --json $O/step_traffic.json > $O/step_traffic.txt; grep "TOTAL per step, every" $O/step_traffic.txt
