#!/bin/bash
# per-kernel times of the head under each ablation build
for n in 0 1 2 8 16 32 59; do
  if [ $n == 0 ]; then unset GFS3D_LIB; else export GFS3D_LIB=$PWD/gfs-3dseg_gws_amd/lib/abl/lib_hd_abl$n.so; fi
  echo "== ABL $n"
  TOPN=10 bash scripts/prof_one.sh scripts/time_capl_head.py 2>&1 | grep "_mm_kernel<16" 
done
