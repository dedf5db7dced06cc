#!/bin/bash
# FETCH_SIZE of the multiply-accumulate kernels at the configs[3] shape (and the headline) for a tuning-knob value:
#   gpurun -- 'bash tools/fetch_ab.sh <tag> <knob> <value> [groth16|rinocchio]'
# one rocprofv3 --pmc pass (kernel-trace only); prints GiB fetched per proof per kernel (x2: the gfx950 correction)
set -u
TAG=$1; KNOB=$2; VAL=$3; WHICH=${4:-rinocchio}
# per-proof figures divide by the proofs ONE prover ran for ONE knob value: both provers emit kernels of the same names
# (mac_kernel_*, plain_*), and every further value runs five more proofs
case "$WHICH" in groth16|rinocchio) ;; *) echo "fetch_ab.sh: which must be groth16 or rinocchio (not '$WHICH': per-proof figures would mix two shapes)"; exit 2;; esac
case "$VAL" in *,*) echo "fetch_ab.sh: one knob value per run (got '$VAL')"; exit 2;; esac
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run --pmc FETCH_SIZE -- python3 tools/knob_ab.py "$KNOB" "$VAL" "$WHICH" > "$OUT/run.log" 2> "$OUT/run.err" || { echo "rocprofv3 failed"; tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg, n = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]); n[k] += 1
proofs = 5.0  # knob_ab.py: 2 warm-up + 3 timed proofs per value
for k in sorted(agg, key=lambda k: -agg[k])[:8]:
    print("%-46s %6.1f launches/proof  fetched %8.2f GiB/proof" % (k[:46], n[k] / proofs, agg[k] * 1024 * 2.0 / proofs / 2**30))
PY
find "$OUT" -name '*.db' -delete
