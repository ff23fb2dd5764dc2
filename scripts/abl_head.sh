#!/bin/bash
# Where the matrix-core passes of the prototype head (csrc/head.hip, hd_*_mm_kernel) spend their time: ablation builds
# (HD_ABL bits: 1 no dot MFMAs, 2 no feature loads, 8 no contraction MFMAs, 16 no d F MFMAs, 32 no d F stores) timed per kernel.
#   bash scripts/abl_head.sh build     (here, no GPU: writes gfs-3dseg_gws_amd/lib/abl/lib_hd_abl<N>.so)
#   bash scripts/abl_head.sh           (on the GPU box: per-kernel times under every build)
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd "$root"
if [ "$1" == "build" ]; then
  for n in 1 2 8 16 32 59; do bash scripts/build_ablation.sh head.hip hd_abl$n -DHD_ABL=$n | tail -1; done
  exit 0
fi
for n in 0 1 2 8 16 32 59; do
  if [ $n == 0 ]; then unset GFS3D_LIB; else export GFS3D_LIB=$root/gfs-3dseg_gws_amd/lib/abl/lib_hd_abl$n.so; fi
  echo "== HD_ABL $n"
  TOPN=12 bash scripts/prof_one.sh scripts/time_capl_head.py 2>&1 | grep "_mm_kernel" || true
done
