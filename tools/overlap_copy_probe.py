"""Does an HBM-bound streaming kernel run UNDER the prover's kernels?  Stream 0: two headline-shape proofs (2^logm
constraints); stream 1: a train of 2 GiB device-to-device copies.  Times: proofs alone, copies alone, both.
(tools/ubench/coresidency.hip is the same question with a synthetic FP64 kernel: there the copy hides under it.)
usage: tools/overlap_copy_probe.py [logm] [logw] [copies]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

logm = int(sys.argv[1]) if len(sys.argv) > 1 else 15
logw = int(sys.argv[2]) if len(sys.argv) > 2 else 11
ncopy = int(sys.argv[3]) if len(sys.argv) > 3 else 60
prm = P.preset("C3")
m, W = 1 << logm, 1 << logw
dev = Device(prm)
dcs = dev.r1cs(R.chain_r1cs(m, prm.q))
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
pk = {k: dev.fill_uniform(dev.enc_empty(W), 1, 13 + i) for i, k in enumerate(("s_pows", "delta_ts", "delta_mid"))}
pk["alpha"], pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 16), dev.fill_uniform(dev.enc_empty(), 1, 17)
src = torch.empty(1 << 28, dtype=torch.int64, device=dev.device).random_()
dst = torch.empty_like(src)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def proofs():
    with torch.cuda.stream(s0):
        for _ in range(2):
            dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W)


def copies():
    with torch.cuda.stream(s1):
        for _ in range(ncopy):
            dst.copy_(src, non_blocking=True)


proofs(); copies(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); proofs(); torch.cuda.synchronize(); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); copies(); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); proofs(); copies(); torch.cuda.synchronize(); tab = time.perf_counter() - t0
    print("2 proofs %.1f ms | %d copies %.1f ms (%.0f GB/s) | together %.1f ms = %.2f of the sum, %.2f of the max" % (
        ta * 1e3, ncopy, tb * 1e3, ncopy * 2 * src.numel() * 8 / tb / 1e9, tab * 1e3, tab / (ta + tb), tab / max(ta, tb)), flush=True)
