#!/bin/bash
# round-2 GPU call 1 (re-entry): where does the stream kernel stand?  correctness, ablations, full GPU suite, bench, kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_1; mkdir -p $O
cd $R
for mode in 3 1; do
  echo "== stream small-shape check (mode $mode)"
  MAS_CONV_STREAM=$mode MAS_CONV_STREAM_MIN_TILES_PER_CU=0 timeout 300 python tests/helpers/stream_check.py 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/stream_check$mode.txt
done
echo "== kbench"
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
{
for act in 0 2; do
  echo -n "old act=$act: "; MAS_CONV_STREAM=0 $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  echo -n "stream act=$act: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  echo -n "imm act=$act: "; MAS_CONV_STREAM=1 $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  for v in noepi nopatch now core nobar; do
    echo -n "$v act=$act: "; MAS_HIP_LIB=$V/s_$v.so $KB conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  done
done
echo -n "res act=2: "; $KB conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 2>&1 | tail -1
echo -n "dgrad: "; $KB dgrad --n 32 --c 128 --hw 256 2>&1 | tail -1
echo -n "dgrad old: "; MAS_CONV_STREAM=0 $KB dgrad --n 32 --c 128 --hw 256 2>&1 | tail -1
for s in "256 64" "512 32" "128 128" "256 128" "512 64"; do set -- $s
  echo -n "c$1 hw$2: "; $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 2>&1 | tail -1
  echo -n "c$1 hw$2 old: "; MAS_CONV_STREAM=0 $KB conv_fwd --n 32 --c $1 --hw $2 --act 2 2>&1 | tail -1
done
for w in 1 2 8; do echo -n "wgs/cu=$w: "; MAS_CONV_WGS_PER_CU=$w $KB conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1; done
for act in 0 2; do echo -n "wgrad act=$act: "; $KB wgrad --n 32 --c 128 --hw 256 --act $act | tail -1; done
} 2>&1 | grep -v amdgpu.ids | tee $O/kbench.txt
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rP --timeout 900 > $O/pytest_full.txt 2>&1; tail -5 $O/pytest_full.txt
echo "== parity prints"
grep -h "img256 bf16\|fwd plain\|fwd GN\|^dgrad:\|wgrad act\|autocast bf16\|^FAILED\|^___" $O/pytest_full.txt | head -40
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-1800 $O/bench.json; tail -3 $O/bench.err
echo "== bench old kernel"
MAS_CONV_STREAM=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err; cut -c1-400 $O/bench_old.json
echo "== trace"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pf_vq -o vq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_vq.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/pf_vq -name "*.db" | head -1) $O/kernel_trace_vq.txt > /dev/null 2>&1; head -40 $O/kernel_trace_vq.txt
