#!/bin/bash
# round 6, call 59: where is the step idle?  (kernel trace of the default build, tools/step_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_59; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace -d /tmp/gp -o gp -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-encoder-stack > /tmp/gp.log 2>&1
python3 $R/tools/step_gaps.py $(find /tmp/gp -name "*.db" | head -1) 40 | tee $O/gaps.txt
