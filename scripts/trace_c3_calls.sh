#!/bin/bash
# Lloyd iterations of the FIRST and of the THIRD c3 flow of one process (scripts/c3_thrice.py), per-kernel: is a slow first call slow on the
# device or on the host?   bash scripts/trace_c3_calls.sh  (on the GPU box)
export TMPDIR=/tmp
d=/tmp/tl_c3c_$$
root=$PWD
(cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 $root/scripts/c3_thrice.py > /tmp/tl_c3c.log 2>&1) || { tail -20 /tmp/tl_c3c.log; exit 1; }
cat /tmp/tl_c3c.log | tail -3
echo "--- first call"; python3 $root/scripts/timeline.py $d/r_results.db km_cent_image 100 dump | sed -n '10,24p'
echo "--- third call"; python3 $root/scripts/timeline.py $d/r_results.db km_cent_image 750 dump | sed -n '10,24p'
python3 $root/scripts/lloyd_gaps.py $d/r_results.db
rm -rf $d
