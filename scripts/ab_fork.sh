#!/bin/bash
# A/B of the reverse-graph fork point of the EdgeConv training forward (csrc/edgeconv.hip: before the projections since round 5) against
# the ablation build with the old one, alternating on ONE box.  Build first, here:  bash scripts/build_ablation.sh edgeconv.hip ec_latefork -DGFS3D_EC_LATE_FORK
one() { if [ "$1" == "old" ]; then export GFS3D_LIB=$PWD/gfs-3dseg_gws_amd/lib/abl/lib_ec_latefork.so; else unset GFS3D_LIB; fi
  python3 bench.py --workload $2 --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'])"; }
for i in 1 2 3; do one new c2; one old c2; done
for i in 1 2; do one new c5; one old c5; done
