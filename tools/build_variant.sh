#!/bin/bash
# build_variant.sh <name> [-DFLAG ...] : builds an experimental libmas_hip variant into make-a-scene_amd/csrc/build/variants/<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/make-a-scene_amd/csrc/build/variants /tmp/var_$name
for f in $R/make-a-scene_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o /tmp/var_$name/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/make-a-scene_amd/csrc/build/variants/$name.so /tmp/var_$name/*.o
echo built $R/make-a-scene_amd/csrc/build/variants/$name.so
