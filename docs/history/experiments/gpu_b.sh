#!/bin/bash
# round-2 GPU call B: stream-kernel correctness, ablations of the stream kernel, full GPU test suite, bench.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/b; mkdir -p $O
cd $R
for mode in 3 1; do
  echo "== stream small-shape check (mode $mode)"
  MAS_CONV_STREAM=$mode MAS_CONV_STREAM_MIN_TILES_PER_CU=0 timeout 300 python tests/helpers/stream_check.py 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/stream_check$mode.txt
done
echo "== kbench ablations"
V=$R/make-a-scene_amd/csrc/build/variants
{
for act in 0 2; do
  echo -n "shipped act=$act: "; timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  for v in noepi nopatch now core nobar corenobar; do
    echo -n "$v act=$act: "; MAS_HIP_LIB=$V/s_$v.so timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act $act 2>&1 | tail -1
  done
done
echo -n "imm act=0: "; MAS_CONV_STREAM=1 timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act 0 2>&1 | tail -1
echo -n "imm act=2: "; MAS_CONV_STREAM=1 timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act 2 2>&1 | tail -1
echo -n "res act=2: "; timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 2>&1 | tail -1
echo -n "dgrad: "; timeout 120 python tools/kbench.py dgrad --n 32 --c 128 --hw 256 2>&1 | tail -1
echo -n "c256: "; timeout 120 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 64 --act 2 2>&1 | tail -1
echo -n "c512: "; timeout 120 python tools/kbench.py conv_fwd --n 32 --c 512 --hw 32 --act 2 2>&1 | tail -1
echo -n "c256 1/CU: "; MAS_CONV_STREAM_MIN_TILES_PER_CU=1 timeout 120 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 32 --act 2 2>&1 | tail -1
echo -n "c256 old: "; MAS_CONV_STREAM=0 timeout 120 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 32 --act 2 2>&1 | tail -1
for w in 1 2 8; do echo -n "wgs/cu=$w: "; MAS_CONV_WGS_PER_CU=$w timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 2>&1 | tail -1; done
} 2>&1 | tee $O/kbench.txt
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rP --timeout 900 > $O/pytest_full.txt 2>&1; tail -5 $O/pytest_full.txt
echo "== parity prints"
grep -h "img256 bf16 vs\|fwd plain\|fwd GN\|^dgrad:\|wgrad act\|^FAILED\|^___" $O/pytest_full.txt | head -40
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
