#!/bin/bash
# Three workgroups per CU for the K4 row GEMMs (csrc/pointwise.hip: PW_RG6_OCC) against the shipped two, per kernel and on the c2 step.
# Build first, here:  bash scripts/build_ablation.sh pointwise.hip pw_occ3 -DPW_RG6_OCC=3      (result: slower, DESIGN_HISTORY)
for rep in 1 2; do
  for v in new old; do
    if [ $v == new ]; then export GFS3D_LIB=$PWD/gfs-3dseg_gws_amd/lib/abl/lib_pw_occ3.so; else unset GFS3D_LIB; fi
    echo "== $v (new = 3 workgroups per CU)"
    TOPN=9 bash scripts/prof_one.sh scripts/time_pw.py 2>&1 | grep "pw_rowgemm6\|pw_wgrad6\|pw_bwd_in"
  done
done
for i in 1 2 3; do
  unset GFS3D_LIB; python3 bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ2 c2', d['ms_per_step'])"
  export GFS3D_LIB=$PWD/gfs-3dseg_gws_amd/lib/abl/lib_pw_occ3.so; python3 bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ3 c2', d['ms_per_step'])"
done
