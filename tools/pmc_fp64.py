"""Summarise the SQ passes of tools/collect_profiles.sh into profiles/<name>.json, keyed by the kernel name rocprofv3
prints (template arguments kept, parameter list dropped):

  usage: tools/pmc_fp64.py <pmc_mix dir> <pmc_stall dir or -> <out.json> <proofs in the run>

Per kernel, PER PROOF (the run's sums divided by the number of prover calls it made: 3 for the collection script's
command): launches, every counter (wave-level instruction counts as rocprofv3 reports them), plus
  fp64_lane_ops  = (SQ_INSTS_VALU_ADD_F64 + _MUL_F64 + _FMA_F64) x 64   -- counted FP64 arithmetic, lane operations
  valu_lane_ops  = SQ_INSTS_VALU x 64                                   -- every vector instruction (v_rndne_f64,
                   moves, integer address arithmetic included: v_rndne_f64 has no opcode counter of its own)
bench.py divides these by the live per-proof time of the same kernel (`roofline.counted`)."""
import collections
import csv
import glob
import json
import sys


def provenance(*dirs):
    """what the counters were collected ON: the hash of the device-library sources of this tree (bench.py refuses to join
    a file whose hash differs from the tree it runs in) and, where git is at hand, the commit"""
    import os
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ringsnark_amd._lib import source_hash
    # tools/collect_profiles.sh records the hash of the tree it profiled next to the passes; without that file (older
    # collections) the hash of the summarising tree is all there is
    recorded = None
    for d in dirs:
        f = os.path.join(os.path.dirname(os.path.normpath(d)), "source_hash.txt")
        if os.path.exists(f):
            recorded = open(f).read().strip()
    out = {"source_hash": recorded or source_hash(), "source_hash_from": "collection" if recorded else "summarising tree"}
    try:
        out["commit"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except OSError:
        out["commit"] = None
    return out


mix, stall, out, proofs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add((f, r["Dispatch_Id"]))
    return agg, {k: len(v) for k, v in calls.items()}


res = {"units": "per proof (one prover call): wave-level counts as reported by rocprofv3; *_lane_ops = x64", "proofs_in_run": proofs, "kernels": {}}
res.update(provenance(mix))
A, nA = load(mix)
B, nB = load(stall) if stall != "-" else ({}, {})
for k in sorted(A, key=lambda k: -A[k].get("SQ_INSTS_VALU", 0)):
    e = {"launches_per_proof": nA[k] / proofs}
    for c, v in sorted(A[k].items()):
        e[c] = v / proofs
    if k in B:
        for c, v in sorted(B[k].items()):
            e[c] = v / proofs
    e["fp64_lane_ops"] = 64.0 * (e.get("SQ_INSTS_VALU_ADD_F64", 0) + e.get("SQ_INSTS_VALU_MUL_F64", 0) + e.get("SQ_INSTS_VALU_FMA_F64", 0))
    e["valu_lane_ops"] = 64.0 * e.get("SQ_INSTS_VALU", 0)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in list(res["kernels"].items())[:12]:
    w = e.get("SQ_WAVE_CYCLES", 0) or 1.0
    print("%-52s %6.1f launches/proof  fp64 %.4g  valu %.4g lane-ops/proof  fp64 share %.2f  wait_any %.2f wait_inst %.2f active_valu %.2f" % (
        k[:52], e["launches_per_proof"], e["fp64_lane_ops"], e["valu_lane_ops"], e["fp64_lane_ops"] / max(1.0, e["valu_lane_ops"]),
        e.get("SQ_WAIT_ANY", 0) / w, e.get("SQ_WAIT_INST_ANY", 0) / w, e.get("SQ_ACTIVE_INST_VALU", 0) / w))
