#!/bin/bash
# The four rocprofv3 passes behind profiles/<tag>_* (run on the GPU box from the repo root):
#   bash scripts/run_profiles.sh <tag>
# kernel trace of the default bench run; SQ counters, FETCH_SIZE and WRITE_SIZE in passes of their own.
set -eo pipefail
tag=${1:?tag}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o r -- python3 bench.py --no-cpu-baseline > "$out/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU \
    -d "$out/sq" -o r -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > "$out/sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o r -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o r -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > "$out/write.log" 2>&1
python3 scripts/collect_profiles.py "$tag" "$out/trace/r_results.db" "$out/sq/r_results.db" "$out/fetch/r_results.db" "$out/write/r_results.db"
mkdir -p "gpurun_out/profiles_$tag" && cp profiles/${tag}_* "gpurun_out/profiles_$tag/"
rm -rf "$out/trace" "$out/sq" "$out/fetch" "$out/write"      # the rocpd databases exceed what gpurun copies back (64 MiB); the summaries are what is kept
grep -h '^{"metric' "$out/trace.log" > "profiles/${tag}_bench_line.json" && cp "profiles/${tag}_bench_line.json" "gpurun_out/profiles_$tag/"
cut -c1-300 "profiles/${tag}_bench_line.json"
