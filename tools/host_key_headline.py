"""The REAL (untiled) key of the 2^16-constraint headline on ONE GPU: 3 x 65536 encoding elements of 2 MiB = 384 GiB
(zk_proof_systems/groth16/groth16.hpp:34-37) do not fit 288 GiB of HBM; they live in page-locked host memory and are
streamed through two device staging buffers under the inner-product kernels (rs_groth16_pk.host_key).  Prints the
time per proof -- bounded by the host link, not by HBM -- and checks the proof against the CPU oracle
(tests/proof_check.py).   usage: tools/host_key_headline.py [logm] [preset] [steps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

logm = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prm = P.preset(sys.argv[2] if len(sys.argv) > 2 else "C3")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = 1 << logm
dev = Device(prm)
key_gib = (3 * m + 2) * prm.enc_words * 8 / 2**30
try:
    limit = int(open("/sys/fs/cgroup/memory.max").read())
except Exception:
    limit = None
avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
print("key %.0f GiB; host memory available %.0f GiB, cgroup limit %s" % (key_gib, avail / 2**30, limit), flush=True)
if key_gib * 2**30 > 0.6 * min(avail, limit or avail):
    print("not enough host memory for the whole key: refusing", flush=True)
    sys.exit(2)
cs = R.chain_r1cs(m, prm.q)
dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
t0 = time.time()
pk = {}
piece = 4096
for i, (name, T) in enumerate((("s_pows", m + 1), ("delta_ts", m + 1), ("delta_mid", cs.n_aux))):
    hw = dev.host_alloc(T * prm.enc_words)
    for a in range(0, T, piece):  # every element distinct: generated on the device piece by piece, copied out
        n = min(piece, T - a)
        hw.fill_from(dev.fill_uniform(dev.enc_empty(n), 1, 1000 * (i + 1) + a), a * prm.enc_words)
    pk[name] = hw
pk["alpha"], pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 16), dev.fill_uniform(dev.enc_empty(), 1, 17)
print("key generated and copied to page-locked host memory in %.1f s" % (time.time() - t0), flush=True)
dev.set_profiling(True)
for s in range(steps):
    t0 = time.time()
    proof, _ = dev.groth16_prove(dcs, pk, asg, want_empty=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("proof %d: %.1f ms = %.0f constraints/s; key streamed at %.1f GB/s; %s" % (s, dt * 1e3, m / dt, key_gib * 2**30 / dt / 1e9, dev.last_timings()), flush=True)
# check two proof slabs against the CPU oracle from the host key itself (A: s_pows, C: delta_ts + delta_mid)
from tests import helpers as H  # noqa: E402
from tests.proof_check import check_columns, slab_inner_product  # noqa: E402
from ringsnark_amd.device import to_host  # noqa: E402
octx = H.oracle_ctx(prm)
w = dev.witness_map(dcs, asg, want=("A_io", "A_mid", "B_io", "B_mid", "H"))
rng = np.random.RandomState(5)
err = check_columns(prm, cs, asg, w, [(int(rng.randint(prm.L)), int(rng.randint(prm.N))) for _ in range(2)], rng)
assert err is None, err
ew = prm.enc_words
slab = lambda hw, T, l, c, j: hw.array.reshape(T, prm.L, 2, prm.K, prm.N_enc)[:, l, c, j, :]
for elem, l, c, j in (("A", 1, 0, 2), ("C", 3, 1, 0)):
    acc = np.zeros(prm.N_enc, dtype=np.uint64)
    if elem == "A":
        k = np.ascontiguousarray(slab(pk["s_pows"], m + 1, l, c, j))
        slab_inner_product(octx, acc, k, w["A_io"], l, j, m)
        slab_inner_product(octx, acc, k, w["A_mid"], l, j, m)
        acc = (acc + to_host(pk["alpha"][l, c, j].contiguous())) % np.uint64(prm.Q[j])
    else:
        slab_inner_product(octx, acc, np.ascontiguousarray(slab(pk["delta_ts"], m + 1, l, c, j)), w["H"], l, j, m + 1)
        slab_inner_product(octx, acc, np.ascontiguousarray(slab(pk["delta_mid"], cs.n_aux, l, c, j)), asg[cs.n_inputs:], l, j, cs.n_aux)
    got = to_host(proof[{"A": 0, "C": 2}[elem], l, c, j].contiguous())
    assert (acc == got).all(), "proof element %s slab differs from the CPU oracle" % elem
print("proof slabs A[limb 1][comp 0][prime 2], C[limb 3][comp 1][prime 0] equal the CPU oracle's; witness identities hold", flush=True)
