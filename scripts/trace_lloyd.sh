#!/bin/bash
# timeline of the Lloyd loop inside scripts/time_c3.py: per iteration (delimited by the assignment kernel) busy vs idle
export TMPDIR=/tmp
d=/tmp/tl_c3_$$
root=$PWD
(cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 $root/scripts/time_c3.py > /tmp/tl_c3.log 2>&1) || { tail -20 /tmp/tl_c3.log; exit 1; }
python3 $root/scripts/timeline.py $d/r_results.db kmeans_assign_x6 150 dump | head -40
rm -rf $d
