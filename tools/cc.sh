#!/bin/bash
# cc.sh <file.hip> [name-substring]: compile one csrc file for gfx950 with resource remarks, keep the ISA in /tmp/isa/, print instruction stats
f=$1; sub=$2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa && cd /tmp/isa
b=$(basename $f .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -save-temps \
  -I$R/make-a-scene_amd/csrc -c $R/make-a-scene_amd/csrc/$b.hip -o /tmp/isa/$b.o 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize" | sed -e 's/.*remark: [^ ]* *//' | paste - - - - - 2>/dev/null | grep "$sub" | cut -c1-250
python3 $R/tools/isa_stats.py /tmp/isa/$b-hip-amdgcn-amd-amdhsa-gfx950.s $sub
