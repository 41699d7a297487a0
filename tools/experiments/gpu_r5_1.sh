#!/bin/bash
# round 5, call 1: the energy budget of the convolution kernels -- MFMA-only loops of >= 5 s with constant / random
# operands, with and without LDS fragment reads, the wide kernel and its ablation builds (built beforehand by tools/build_file_variant.sh:
# wabl_nopatch, wabl_now, wabl_noepi, wabl_mfma, wabl_halfa, wabl_mfma_halfa), clock and package power sampled over the steady state;
# then who issues the copyBuffer launches of a no_grad Encoder forward, and the default bench line of the round-4 build on this box.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r5_1; mkdir -p $O
[ -x tools/probes/mfma_power ] || bash tools/probes/build.sh > /dev/null 2>&1
timeout 900 python tools/probes/energy_budget.py 5 > $O/energy_budget.txt 2>&1; cut -c1-250 $O/energy_budget.txt
timeout 300 python tools/probes/op_origin.py --mode enc_fwd --ops aten::copy_,aten::_to_copy,aten::fill_,aten::zero_,aten::cat,aten::contiguous > $O/op_origin_enc.txt 2>&1; tail -40 $O/op_origin_enc.txt | cut -c1-220
timeout 600 python bench.py --no-cpu-baseline --no-also > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-1500
