for n in 0 1 2 3 4 5 7; do
  if [ $n == 0 ]; then unset GFS3D_LIB; else export GFS3D_LIB=$PWD/gfs-3dseg_gws_amd/lib/abl/lib_ec_abl$n.so; fi
  echo "ABL $n"; timeout -k 10 120 python scripts/exp_reduced.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['mode'], 'eval', d['eval_fwd_ms'], 'train_fwd', d['train_fwd_ms'], 'bwd', d['bwd_ms'])"
done
