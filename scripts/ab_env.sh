#!/bin/bash
# A/B of one library under an environment switch on ONE box: bash scripts/ab_env.sh "<VAR=value>" [workload] [rounds]
# prints ms per step without ("new") and with ("old") the variable set, alternating.
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
var=$1; wl=${2:-c2}; rounds=${3:-3}
log=$(mktemp /tmp/ab_env.XXXXXX)
one() {
  if ! out=$(env $2 python3 "$root/bench.py" --workload $wl --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>"$log"); then
    echo "$1: bench.py failed:" >&2; tail -20 "$log" >&2; return 1
  fi
  echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"
}
for i in $(seq $rounds); do
  one new "GFS3D_AB_NONE=1"
  one old "$var"
done
