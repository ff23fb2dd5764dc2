#!/bin/bash
# The four rocprofv3 passes behind profiles/<tag>_<workload>_* (run on the GPU box from the repo root):
#   bash scripts/run_profiles.sh <tag> [c2|c4|c5|all]
# kernel trace of a bench run; SQ counters, FETCH_SIZE and WRITE_SIZE in passes of their own.  c2 / c4 / c5: that workload alone
# (--no-extras; `_steps` in the traffic summary = warm-up + timed + instrumented steps, so that bench.py can turn the bytes of a
# kernel GROUP into bytes per step); all: the default bench with its side measurements (Lloyd, seeding, heads, attention).
set -eo pipefail
tag=${1:?tag}
wl=${2:-c2}
export TMPDIR=/tmp
out=gpurun_out/prof_${tag}_$wl
mkdir -p "$out"
if [ "$wl" = all ]; then args="--no-cpu-baseline"; pargs="--no-cpu-baseline --steps 3 --warmup 2"; steps=0
else args="--workload $wl --no-extras --no-cpu-baseline"; pargs="$args --steps 3 --warmup 2"; steps=8; fi
rocprofv3 --kernel-trace --stats -d "$out/trace" -o r -- python3 bench.py $args > "$out/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU \
    -d "$out/sq" -o r -- python3 bench.py $pargs > "$out/sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o r -- python3 bench.py $pargs > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o r -- python3 bench.py $pargs > "$out/write.log" 2>&1
python3 scripts/collect_profiles.py "${tag}_$wl" "$out/trace/r_results.db" "$out/sq/r_results.db" "$out/fetch/r_results.db" "$out/write/r_results.db" \
    "$steps" "python3 bench.py $pargs"
mkdir -p "gpurun_out/profiles_${tag}_$wl" && cp profiles/${tag}_${wl}_* "gpurun_out/profiles_${tag}_$wl/"
rm -rf "$out/trace" "$out/sq" "$out/fetch" "$out/write"      # the rocpd databases exceed what gpurun copies back (64 MiB); the summaries are what is kept
grep -h '^{"metric' "$out/trace.log" > "profiles/${tag}_${wl}_bench_line.json" && cp "profiles/${tag}_${wl}_bench_line.json" "gpurun_out/profiles_${tag}_$wl/"
cut -c1-300 "profiles/${tag}_${wl}_bench_line.json"
