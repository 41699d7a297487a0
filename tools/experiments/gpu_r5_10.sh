#!/bin/bash
# round 5, call 10: block budget of gn_act / gn_bwd apply in the step (same-box A/B, two repetitions), and the per-kernel split of gn_bwd at the cold size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_10; mkdir -p $O
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "4096 4096" "16384 8192" "16384 16384" "8192 8192"; do
  set -- $cfg
  echo "== MAS_GN_ACT_BLOCKS=$1 MAS_GN_APPLY_BLOCKS=$2 (rep $rep)"
  MAS_GN_ACT_BLOCKS=$1 MAS_GN_APPLY_BLOCKS=$2 timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-also --no-encoder-stack 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done; done 2>&1 | tee $O/gn_blocks_step.txt
for cfg in "4096 4096" "16384 8192"; do
  set -- $cfg
  rm -rf /tmp/pf_gn
  cd /tmp && MAS_GN_ACT_BLOCKS=$1 MAS_GN_APPLY_BLOCKS=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_gn -o gn -- python $R/tools/probes/gn_blocks_probe.py child > /tmp/pf_gn.log 2>&1
  cd $R && python tools/rocprof_summary.py $(find /tmp/pf_gn -name "*.db" | head -1) $O/gn_probe_trace_$1_$2.txt > /dev/null 2>&1
  echo "== trace act=$1 apply=$2"; grep -i "gn_\|elementwise" $O/gn_probe_trace_$1_$2.txt | cut -c1-170 | head -14
done
