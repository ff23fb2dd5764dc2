#!/bin/bash
# Timeline of any script's launches between consecutive hits of a marker kernel: bash scripts/trace_script.sh <script.py> <marker substring> [skip] [dump]
# (scripts/timeline.py on a rocprofv3 --kernel-trace of the script: per window launches / busy / idle, one window listed in start order)
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
d=/tmp/trace_script_$$
log=/tmp/trace_script_$$.log
if ! (cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 "$root/$1" > $log 2>&1); then
  echo "trace_script: rocprofv3 pass failed, log follows" >&2; tail -30 $log >&2; exit 1
fi
python3 "$root/scripts/timeline.py" $d/r_results.db "$2" "${3:-4}" ${4:-}
rm -rf $d $log
