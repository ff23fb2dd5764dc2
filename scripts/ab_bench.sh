#!/bin/bash
# A/B of two builds of the library on ONE box (devices differ by several per cent): bash scripts/ab_bench.sh <other lib .so> [workload] [rounds]
# prints ms per step of the in-tree library ("new") and of the other one, alternating.
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
other=$1; wl=${2:-c2}; rounds=${3:-3}
log=$(mktemp /tmp/ab_bench.XXXXXX)
one() {
  # stderr is kept: a run that fails (e.g. a GFS3D_LIB that does not load) shows its real error, not a JSON parser traceback
  if ! out=$(python3 "$root/bench.py" --workload $wl --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>"$log"); then
    echo "$1: bench.py failed:" >&2; tail -20 "$log" >&2; return 1
  fi
  echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"
}
for i in $(seq $rounds); do
  one new
  GFS3D_LIB=$other one old
done
