#!/bin/bash
# round 4, session 13: who issues the ~270 fill and ~100 copy launches of a training step (tools/probes/op_origin.py --mode step)
cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_13; mkdir -p $O
timeout 300 python tools/probes/op_origin.py --mode step --ops aten::fill_,aten::zero_,aten::copy_,aten::cat,aten::add,aten::add_ 2>&1 | grep -v Warning | cut -c1-330 > $O/op_origin_step.txt; head -45 $O/op_origin_step.txt
