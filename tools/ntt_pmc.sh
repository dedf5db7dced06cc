#!/bin/bash
# SQ stall / instruction-mix counters of the standalone wide transforms (tools/ntt16_pmc.py) at 16384 (preset C4) and 8192
# (preset C3) points.  PMC passes with --kernel-trace only.  usage: gpurun -- 'bash tools/ntt_pmc.sh [tag]'
set -eu
TAG=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for P in C4 C3; do
  OUT="gpurun_out/${TAG}_ntt_pmc_$P"
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/stall" -o run --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python3 tools/ntt16_pmc.py $P 1 3 > "$OUT.log" 2>&1
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/mix" -o run --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -- python3 tools/ntt16_pmc.py $P 1 3 >> "$OUT.log" 2>&1
  find "$OUT" -name '*.db' -delete
done
ls -R "gpurun_out/${TAG}_ntt_pmc_C4" | head
