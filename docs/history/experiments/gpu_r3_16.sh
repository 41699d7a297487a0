#!/bin/bash
# round-3 GPU call 19: attention with scalar (un-packed) softmax VALU / without SLP packing: kbench + transformer step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
KB="timeout 120 python tools/kbench.py"
for v in main scalar noslp main scalar noslp; do
  if [ $v = main ]; then L="X=1"; else L="MAS_HIP_LIB=$V/attn_$v.so"; fi
  echo "== kbench attn [$v]"
  for n in 8 16; do env $L $KB attn --n $n 2>&1 | grep -v amdgpu.ids | head -2; done
done
echo "== parity (scalar variant)"
MAS_HIP_LIB=$V/attn_scalar.so timeout 600 python -m pytest tests/test_gpu_transformer.py -m gpu -q -k "attention" 2>&1 | tail -2
