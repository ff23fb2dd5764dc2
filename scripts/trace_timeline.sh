#!/bin/bash
# Timeline summary (busy / idle per step) of one bench workload: bash scripts/trace_timeline.sh <c2|c4|c5> <out.txt>
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
wl=$1; out=$2
d=/tmp/trace_tl_$$
log=/tmp/trace_tl_$$.log
if ! (cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 "$root/bench.py" --workload $wl --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $log 2>&1); then
  echo "trace_timeline: rocprofv3 pass failed, log follows" >&2; tail -40 $log >&2; exit 1
fi
python3 "$root/scripts/timeline.py" $d/r_results.db multi_tensor_apply 4 ${3:-} > "$out"
rm -rf $d $log
