#!/bin/bash
# Extra SQ counter passes over the default bench (run on the GPU box): bash scripts/pmc_extra.sh <out.txt> [filter]
# Counter passes only (no trace domains beside --pmc); a failed pass stops the script and shows the bench's log.
set -eo pipefail
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export TMPDIR=/tmp
out=$1; filt=${2:-}
d=/tmp/pmcx_$$
log=/tmp/pmcx_$$.log
: > "$out"
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  rm -rf $d
  if ! (cd /tmp && rocprofv3 --pmc $set -d $d -o r -- python3 "$root/bench.py" --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $log 2>&1); then
    echo "pmc_extra: rocprofv3 pass failed, log follows" >&2; tail -40 $log >&2; exit 1
  fi
  python3 "$root/scripts/prof_pmc.py" $d/r_results.db "$filt" >> "$out"
done
rm -rf $d $log
