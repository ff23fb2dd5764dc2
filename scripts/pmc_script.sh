#!/bin/bash
# SQ / LDS / VMEM counter passes over ONE script (run on the GPU box): bash scripts/pmc_script.sh <out.txt> <kernel filter> <script.py> [args]
# Three rocprofv3 --pmc passes (counters only, no trace domains), summed per kernel by scripts/prof_pmc.py.
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
out=$1; filt=$2; shift 2
d=/tmp/pmcs_$$
log=/tmp/pmcs_$$.log
: > "$out"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $d
  if ! (cd /tmp && rocprofv3 --pmc $set -d $d -o r -- python3 "$root/$1" "${@:2}" > $log 2>&1); then
    echo "pmc_script: rocprofv3 pass failed, log follows" >&2; tail -40 $log >&2; exit 1
  fi
  python3 "$root/scripts/prof_pmc.py" $d/r_results.db "$filt" >> "$out"
done
rm -rf $d $log
