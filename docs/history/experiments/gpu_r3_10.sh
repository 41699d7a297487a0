#!/bin/bash
# round-3 GPU call 13: why is MAS_OVERLAP=1 slower?  kernel timelines (queue ids) of one ResnetBlock backward, with and without
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3_10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ov in 1 0; do
  rm -rf /tmp/pf_ov
  MAS_OVERLAP=$ov MAS_OVERLAP_GN_CUS=112 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_ov -o ov -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also > /tmp/pf_ov.log 2>&1
  echo "== MAS_OVERLAP=$ov"; tail -1 /tmp/pf_ov.log | cut -c1-200
  python $R/tools/overlap_timeline.py $(find /tmp/pf_ov -name "*.db" | head -1) | tee $O/timeline_ov$ov.txt
done
