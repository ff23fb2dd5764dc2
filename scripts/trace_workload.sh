#!/bin/bash
# Kernel trace of ONE bench workload without the side measurements: bash scripts/trace_workload.sh <c2|c4|c5> <out.txt> [steps]
# -> per-kernel totals of the traced process (raw names), for a per-step breakdown of that workload alone.
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
wl=$1; out=$2; steps=${3:-10}
d=/tmp/trace_wl_$$
log=/tmp/trace_wl_$$.log
if ! (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 "$root/bench.py" --workload $wl --no-extras --no-cpu-baseline --steps $steps --warmup 2 > $log 2>&1); then
  echo "trace_workload: rocprofv3 pass failed, log follows" >&2; tail -40 $log >&2; exit 1
fi
grep -h '^{"metric' $log | cut -c1-400 > "$out"
python3 "$root/scripts/prof_summary.py" $d/r_results.db 400 >> "$out"
rm -rf $d $log
