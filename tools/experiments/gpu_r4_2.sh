#!/bin/bash
# round 4, session 2: full GPU suite after the dispatch changes (materialised GroupNorm+SiLU also under no_grad, side output removed),
# default bench line (encoder_stack, peak memory), copy / fill origins of the no_grad Encoder forward
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 12 --no-cpu-baseline --no-also > $O/bench_vq.json 2> $O/bench_vq.err; tail -c 1500 $O/bench_vq.json
timeout 200 python tools/probes/op_origin.py --mode enc_fwd > $O/op_origin_enc.txt 2>&1; tail -25 $O/op_origin_enc.txt
timeout 200 python tools/probes/op_origin.py --mode step > $O/op_origin_step.txt 2>&1; tail -25 $O/op_origin_step.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_enc -o enc -- python $GRAFT_REPO_ROOT/tools/enc_fwd.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $(find /tmp/pf_enc -name "*.db" | head -1) $O/kernel_trace_encoder_fwd.txt > /dev/null 2>&1; head -30 $O/kernel_trace_encoder_fwd.txt | cut -c1-180
