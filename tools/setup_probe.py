"""Where the first proof of a (context, m) spends its extra time: the witness-map plan (host tables), the first-call
workspace allocations, the io-vector cache.  usage: gpurun -- python tools/setup_probe.py [preset=C3] [logm=16]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from ringsnark_amd import params as P, r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

prm = P.preset(sys.argv[1] if len(sys.argv) > 1 else "C3")
m = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print("%-58s %8.1f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
    return r


dev = timed("context", lambda: Device(prm))
cs = R.chain_r1cs(m, prm.q)
dcs = timed("r1cs upload", lambda: dev.r1cs(cs))
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
W = 1 << 12
pk = {k: dev.fill_uniform(dev.enc_empty(W), 1, 13 + i) for i, k in enumerate(("s_pows", "delta_ts", "delta_mid"))}
pk["alpha"], pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 16), dev.fill_uniform(dev.enc_empty(), 1, 17)
timed("witness map on 2 slots (builds the plan + io cache)", lambda: dev.witness_map_slots(dcs, asg, 0, 2, want=("A_mid", "B_mid", "H")))
timed("witness map on 2 slots again", lambda: dev.witness_map_slots(dcs, asg, 0, 2, want=("A_mid", "B_mid", "H")))
timed("first proof (workspace allocations)", lambda: dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W))
timed("second proof", lambda: dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W))
timed("third proof", lambda: dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W))
