#!/bin/bash
# Ablation builds of the library: bash scripts/build_ablation.sh <source.hip> <name> <extra hipcc flags...>
# -> gfs-3dseg_gws_amd/lib/abl/lib_<name>.so (the other objects come from lib/obj); run with GFS3D_LIB=<that path>.
set -eo pipefail
src=$1; name=$2; shift 2
pkg=$(dirname "$0")/../gfs-3dseg_gws_amd
mkdir -p "$pkg/lib/abl"
flags="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value"
/opt/rocm/bin/hipcc $flags "$@" -c "$pkg/csrc/$src" -o "$pkg/lib/abl/$name.o"
objs=""
for f in api knn edgeconv proto pointwise attention head kmeanspp optim; do
  if [ "$f.hip" == "$src" ]; then objs="$objs $pkg/lib/abl/$name.o"; else objs="$objs $pkg/lib/obj/$f.hip.o"; fi
done
/opt/rocm/bin/hipcc $flags -shared $objs -o "$pkg/lib/abl/lib_$name.so"
rm -f "$pkg/lib/abl/$name.o"
echo "$pkg/lib/abl/lib_$name.so"
