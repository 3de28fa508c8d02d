#!/bin/bash
# rocprofv3 counter passes (one --pmc set per run, kernel trace only) around a python probe; CSVs under $OUT.
#   tools/pmc_run.sh <outdir> <python script> [args...]      (environment is passed through)
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/errors.txt"
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_MISC
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
FETCH_SIZE
WRITE_SIZE
SETS
