#!/bin/bash
# Collects, on the GPU box, every rocprofv3 record the bench line's rooflines are recomputed from:
#   1. kernel trace + stats of the default bench command (NTT on 4 GiB included)
#   2. SQ instruction-mix pass (FP64 opcode counters) and SQ stall pass of one headline proof
#   3. FETCH_SIZE and WRITE_SIZE passes (separate: they do not fit one pass)
# PMC passes run with --kernel-trace only (never with other trace domains).  Outputs land under gpurun_out/<tag>/;
# tools/pmc_fp64.py and tools/pmc_summary.py turn them into the files kept under profiles/.
#   usage: gpurun -- 'bash tools/collect_profiles.sh <tag> [preset] [stages]'    stages: subset of "kpst" (default all)
set -u
TAG=${1:-r04}
PRESET=${2:-C3}
STAGES=${3:-kpst}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
# what is being profiled: the hash of the device-library sources of THIS tree (bench.py joins a counter file only when it matches)
python3 -c "from ringsnark_amd._lib import source_hash; print(source_hash())" > "$OUT/source_hash.txt"
# a failed pass (counter set rejected, bench error) must not leave partial csv files for the summarisers (round-3 advice)
FAILED=0
run_pass() {  # run_pass <name> <expected csv glob> <command...>
  local name=$1 want=$2
  shift 2
  "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  local rc=$?
  if [[ $rc -ne 0 ]]; then echo "collect_profiles: pass $name FAILED (rc=$rc); see $OUT/$name.err" >&2; FAILED=1; return; fi
  if ! compgen -G "$OUT/$want" > /dev/null || [[ ! -s $(compgen -G "$OUT/$want" | head -1) ]]; then
    echo "collect_profiles: pass $name wrote no $want" >&2; FAILED=1
  fi
}
PROBE="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-ntt --no-recipe-primes --no-other-configs --preset $PRESET"  # exactly 3 proofs: warm-up, timed, profiled
if [[ $STAGES == *k* ]]; then
  run_pass bench_stats "stats/*kernel_stats.csv" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-recipe-primes --no-other-configs --preset "$PRESET"
fi
if [[ $STAGES == *p* ]]; then
  run_pass pmc_mix "pmc_mix/*counter_collection.csv" rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_mix" -o run \
    --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SALU \
    -- $PROBE
fi
if [[ $STAGES == *s* ]]; then
  run_pass pmc_stall "pmc_stall/*counter_collection.csv" rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_stall" -o run \
    --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    -- $PROBE
fi
if [[ $STAGES == *t* ]]; then
  run_pass pmc_fetch "pmc_fetch/*counter_collection.csv" rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run --pmc FETCH_SIZE -- $PROBE
  run_pass pmc_write "pmc_write/*counter_collection.csv" rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run --pmc WRITE_SIZE -- $PROBE
fi
# keep what travels back small: the per-dispatch csv files are what the summarisers read
if [[ $FAILED -ne 0 ]]; then echo "collect_profiles: at least one pass failed; nothing deleted, do not summarise" >&2; exit 1; fi
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
