#!/bin/bash
# build_file_variant.sh <name> <file.hip> [-DFLAG ...]: like build_variant.sh but recompiles ONE csrc file with the flags and links it
# with the already-built objects of the others (make-a-scene_amd/csrc/build/*.o) -> make-a-scene_amd/csrc/build/variants/<name>.so
set -e
name=$1; file=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/make-a-scene_amd/csrc/build
mkdir -p $B/variants /tmp/var_$name
b=$(basename $file .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $R/make-a-scene_amd/csrc/$b.hip -o /tmp/var_$name/$b.o
objs=""
for o in $B/*.o; do [ "$(basename $o .o)" = "$b" ] && objs="$objs /tmp/var_$name/$b.o" || objs="$objs $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/variants/$name.so $objs
echo built $B/variants/$name.so
