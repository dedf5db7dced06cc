"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace
only) of `bench.py --steps 1 --warmup 1` into profiles/<name>.json.

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled.
Calibration inside the same run: transpose_out_kernel reads S*M*8 bytes and writes m*S*8 bytes of
known size -- the summary records reported vs expected for both."""
import csv, glob, json, sys, collections

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


fdir, wdir, out, proofs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
def load(d, c):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg
F, W = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
res = {"units": "bytes per proof (one prover call)", "proofs_in_run": proofs, "fetch_correction": 2.0, "kernels": {}}
res.update(provenance(fdir))
for k in sorted(set(F) | set(W)):
    n = max(len(F.get(k, [])), len(W.get(k, [])))
    fetch = sum(F.get(k, [])) * 1024 * 2.0 / proofs
    write = sum(W.get(k, [])) * 1024 / proofs
    res["kernels"][k] = {"launches_per_proof": n / proofs, "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write}
t = res["kernels"].get("rs::transpose_out_kernel")
if t:
    per = t["launches_per_proof"]
    res["calibration"] = {"kernel": "rs::transpose_out_kernel", "reported_fetch_per_launch_raw": t["fetch_bytes"] / 2 / per,
                          "reported_write_per_launch": t["write_bytes"] / per,
                          "note": "reads and writes 8 B per element of an [m][L*N] vector: expected 2 GiB each at C3, m=2^13"}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes"])[:8]:
    print("%-40s launches/proof %5.1f  fetch %8.2f GiB  write %7.2f GiB" % (k, v["launches_per_proof"], v["fetch_bytes"] / 2**30, v["write_bytes"] / 2**30))
