#!/bin/bash
# round 6, call 2: the new / changed tests (BatchNorm fp64 sums + eval autograd + fused LeakyReLU, discriminator, benched-size VQ golden, B=32
# model-level parity, fp32 model with bf16 spatial attention, colsum hand-off, UP2 slices)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r6_2; mkdir -p $O
timeout 1500 python -m pytest -m gpu -q -rP --timeout 900 tests/test_gpu_bn.py tests/test_gpu_losses.py tests/test_gpu_transformer.py tests/test_gpu_up2.py \
  "tests/test_gpu_kernels.py::test_vq_lookup_at_the_benched_size_vs_reference_golden" "tests/test_gpu_kernels.py::test_vq_lookup_vs_reference_golden" \
  "tests/test_gpu_model.py::test_img256_fp32_with_bf16_spatial_attention_vs_reference_golden" \
  "tests/test_gpu_parity_r3.py::test_img256_bf16_multi_tile_batch_vs_oracle" tests/test_gpu_parity_r2.py --durations=8 > $O/pytest.txt 2>&1
grep -E "passed|failed|error|index agreement|codebook_b32|spatial attention vs|^[0-9.]+s " $O/pytest.txt | tail -40
