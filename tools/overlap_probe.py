"""Do two independent provers overlap on one GPU?  Two contexts, two HIP streams, two host threads, the headline shape at
m = 2^logm each (tiled key): total time of two proofs back to back on one stream against two proofs issued concurrently.
If the HBM-bound passes of one fill the VALU-idle time of the other, the concurrent run is shorter than the serial one.
usage: tools/overlap_probe.py [logm] [logw] [preset]"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

logm = int(sys.argv[1]) if len(sys.argv) > 1 else 15
logw = int(sys.argv[2]) if len(sys.argv) > 2 else 11
prm = P.preset(sys.argv[3] if len(sys.argv) > 3 else "C3")
m, W = 1 << logm, 1 << logw


def setup():
    dev = Device(prm)
    cs = R.chain_r1cs(m, prm.q)
    dcs = dev.r1cs(cs)
    asg = dev.ring_empty(m + 2)
    dev.fill_uniform(asg[:2], 0, 7)
    dev.chain_assignment(asg, m)
    pk = {k: dev.fill_uniform(dev.enc_empty(W), 1, 13 + i) for i, k in enumerate(("s_pows", "delta_ts", "delta_mid"))}
    pk["alpha"], pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 16), dev.fill_uniform(dev.enc_empty(), 1, 17)
    return dev, dcs, asg, pk


A, B = setup(), setup()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def prove(ctx, stream, n):
    dev, dcs, asg, pk = ctx
    with torch.cuda.stream(stream):
        for _ in range(n):
            dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W)


for ctx, s in ((A, streams[0]), (B, streams[1])):  # warm-up: plans, io cache
    prove(ctx, s, 1)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    prove(A, streams[0], 2)
    prove(B, streams[0], 2)
    torch.cuda.synchronize()
    serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=prove, args=(A, streams[0], 2)), threading.Thread(target=prove, args=(B, streams[1], 2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    conc = time.perf_counter() - t0
    print("4 proofs of 2^%d constraints: one stream %.1f ms, two streams %.1f ms (%.3f)" % (logm, serial * 1e3, conc * 1e3, conc / serial), flush=True)
