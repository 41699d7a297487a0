#!/bin/bash
# round-2 GPU call A: DMA probe, full GPU test suite, stream-kernel A/B micro-benchmarks, bench.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/a; mkdir -p $O
cd $R
echo "== probe"; timeout 60 tools/probes/dma_oob_probe 2>&1 | tee $O/probe.txt
echo "== stream small-shape check (mode 3)"
MAS_CONV_STREAM=3 MAS_CONV_STREAM_MIN_TILES_PER_CU=0 timeout 300 python tests/helpers/stream_check.py 2>&1 | tail -15 | tee $O/stream_check3.txt
echo "== kbench"
{
for mode in 0 1 3; do
  for act in 0 2; do
    echo -n "STREAM=$mode "; MAS_CONV_STREAM=$mode timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act $act | tail -1
  done
  echo -n "STREAM=$mode res "; MAS_CONV_STREAM=$mode timeout 120 python tools/kbench.py conv_fwd --n 32 --c 128 --hw 256 --act 2 --res 1 | tail -1
  echo -n "STREAM=$mode "; MAS_CONV_STREAM=$mode timeout 120 python tools/kbench.py dgrad --n 32 --c 128 --hw 256 | tail -1
  echo -n "STREAM=$mode "; MAS_CONV_STREAM=$mode timeout 120 python tools/kbench.py conv_fwd --n 32 --c 256 --hw 64 --act 2 | tail -1
  echo -n "STREAM=$mode "; MAS_CONV_STREAM=$mode timeout 120 python tools/kbench.py conv_fwd --n 32 --c 512 --hw 32 --act 2 | tail -1
done
} 2>&1 | tee $O/kbench.txt
echo "== pytest"
timeout 1500 python -m pytest tests -m gpu -q -rP --timeout 900 > $O/pytest_full.txt 2>&1; tail -15 $O/pytest_full.txt
echo "== parity prints"
grep -h "img256 bf16\|fwd plain\|fwd GN\|^dgrad:\|wgrad act\|autocast bf16\|  grad \|FAILED\|Error" $O/pytest_full.txt | head -40
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
