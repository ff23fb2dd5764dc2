#!/bin/bash
# timeline of the Lloyd loop inside scripts/c3_only.py: per iteration (delimited by the image kernel of the screened assignment) busy vs idle
export TMPDIR=/tmp
d=/tmp/tl_c3_$$
root=$PWD
(cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 $root/scripts/c3_only.py > /tmp/tl_c3.log 2>&1) || { tail -20 /tmp/tl_c3.log; exit 1; }
python3 $root/scripts/timeline.py $d/r_results.db ${1:-km_cent_image} 150 dump | head -40
rm -rf $d
