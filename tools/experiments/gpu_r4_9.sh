#!/bin/bash
# round 4, session 9: spatial attention phase timing (-DSP_TRACE variant) and parity of the register-resident rewrite
cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spatial_attn.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-200 > $O/pytest.txt; cat $O/pytest.txt
{ python tools/kbench.py sp_attn --iters 50 2>&1 | grep sp_attn
python tools/kbench.py sp_attn --hw 8 --iters 50 2>&1 | grep sp_attn
MAS_HIP_LIB=make-a-scene_amd/csrc/build/variants/sptrace.so python tools/kbench.py sp_attn --iters 50 2>&1 | grep -A2 "sp_attn forward phases"; } > $O/kbench_sp.txt 2>&1; cat $O/kbench_sp.txt
