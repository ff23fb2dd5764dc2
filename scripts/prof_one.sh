#!/bin/bash
# rocprofv3 kernel stats of one script: bash scripts/prof_one.sh <script.py> [args]  (prints the top kernels)
# The repo root is derived from this file's location; the profiled program's output is kept and shown when the pass fails.
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
d=/tmp/prof_one_$$
log=/tmp/prof_one_$$.log
if ! (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 "$root/$1" "${@:2}" > $log 2>&1); then
  echo "prof_one: rocprofv3 pass failed, log follows" >&2; tail -40 $log >&2; exit 1
fi
python3 "$root/scripts/prof_summary.py" $d/r_results.db | (head -${TOPN:-12} || true) | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r'calls=.*', l); name=re.sub(r'\\(.*','',l.split('calls=')[0].replace('(anonymous namespace)::','').replace('void ','')).strip()
    print(name[:60].ljust(62), m.group(0) if m else '')"
rm -rf $d $log
