#!/bin/bash
# Instruction-cache and LDS-queue counters of one headline proof (two --pmc passes, kernel trace only):
#   gpurun -- 'bash tools/icache_pmc.sh <tag>'   -> gpurun_out/<tag>/pmc_icache, pmc_lds ; summarise with tools/pmc_table.py
TAG=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
PROBE="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-ntt --no-recipe-primes --no-other-configs --preset C3"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_icache" -o run \
  --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  -- $PROBE > "$OUT/pmc_icache.json" 2> "$OUT/pmc_icache.err"; echo icache rc=$?
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_lds" -o run \
  --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
  -- $PROBE > "$OUT/pmc_lds.json" 2> "$OUT/pmc_lds.err"; echo lds rc=$?
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
