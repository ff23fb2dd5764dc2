#!/bin/bash
# rocprofv3 kernel stats of one script, raw kernel names: bash scripts/prof_raw.sh <out.txt> <script.py> [args]
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
d=/tmp/prof_raw_$$
log=/tmp/prof_raw_$$.log
if ! (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 "$root/$2" "${@:3}" > $log 2>&1); then
  echo "prof_raw: rocprofv3 pass failed, log follows" >&2; tail -40 $log >&2; exit 1
fi
python3 "$root/scripts/prof_summary.py" $d/r_results.db 400 > "$1"
rm -rf $d $log
