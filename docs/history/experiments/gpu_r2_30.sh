#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$R/make-a-scene_amd/csrc/build/variants
MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 0 2>&1 | grep -v amdgpu.ids
MAS_HIP_LIB=$V/wg_tl.so timeout 120 python tools/timeline_wgrad.py 2 2>&1 | grep -v amdgpu.ids
