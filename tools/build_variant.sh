#!/bin/bash
# Developer A/B: a second libtfgx.so with ONE source recompiled under extra flags (the other objects are reused).
#   tools/build_variant.sh <name> <source.hip> "<flags>"   ->  tf_geometric_amd/lib/variants/<name>/libtfgx.so
# Load it with TFGX_LIB_PATH=tf_geometric_amd/lib/variants/<name>/libtfgx.so (tf_geometric_amd/_lib.py).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; src=$2; flags=$3
dir="$ROOT/tf_geometric_amd/lib/variants/$name"
mkdir -p "$dir"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c "$ROOT/tf_geometric_amd/csrc/$src" -o "$dir/${src%.hip}.o"
objs=""
for o in "$ROOT"/tf_geometric_amd/lib/obj/*.o; do
  b=$(basename "$o")
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs $dir/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$dir/libtfgx.so" $objs
echo "$dir/libtfgx.so"
