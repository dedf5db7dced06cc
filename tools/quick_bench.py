"""Ad-hoc kernel timings on the GPU box (not the contract bench; see bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, r1cs as R, _lib
from ringsnark_amd.device import Device

def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
logm = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prm = P.preset(name)
dev = Device(prm)
m = 1 << logm
print("preset", name, "N", prm.N, "L", prm.L, "N_enc", prm.N_enc, "K", prm.K, "m", m, flush=True)
# NTT
batch = 4096
d = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device)
dev.fill_uniform(d.view(-1, prm.L, prm.N)[: batch * prm.N_enc // prm.ring_words], 0, 1) if False else None
d.random_(0, prm.Q[0])
ms = timeit(lambda: dev.ntt(d, _lib.RS_MOD_COEFF, 0))
gb = batch * prm.N_enc * 16 / 1e9
print("ntt fwd: %.3f ms for %d polys -> %.1f GB/s algorithmic, %.2f us/NTT" % (ms, batch, gb / ms * 1e3, ms * 1e3 / batch), flush=True)
ms = timeit(lambda: dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=True))
print("ntt inv: %.3f ms -> %.1f GB/s" % (ms, gb / ms * 1e3), flush=True)
# ring mul
a = dev.ring_empty(2048); b = dev.ring_empty(2048)
dev.fill_uniform(a, 0, 1); dev.fill_uniform(b, 0, 2)
ms = timeit(lambda: dev.ring_mul(a, b))
print("ring_mul: %.3f ms -> %.1f GB/s" % (ms, a.numel() * 24 / 1e6 / ms), flush=True)
# MSM (single inner product and grouped)
T = m
crs = dev.enc_empty(T + 1); dev.fill_uniform(crs, 1, 3)
v = [dev.ring_empty(T) for _ in range(4)]
for k, x in enumerate(v): dev.fill_uniform(x, 0, 10 + k)
dev.set_profiling(True)
ms = timeit(lambda: dev.inner_product(crs[:T], v[0], want_used=False), reps=3, warm=1)
ctb = T * prm.enc_words * 8
print("inner_product T=%d: %.3f ms, %.2f us/term, ct stream %.1f GB/s" % (T, ms, ms * 1e3 / T, ctb / 1e6 / ms), flush=True)
ms = timeit(lambda: dev.msm([crs[:T]], [(v[0], None, 0), (v[1], None, 0), (v[2], None, 1), (v[3], None, 1)], 2), reps=3, warm=1)
print("grouped msm (2 groups x 2 vecs) T=%d: %.3f ms, %.2f us/term, ct stream %.1f GB/s" % (T, ms, ms * 1e3 / T, ctb / 1e6 / ms), flush=True)
# witness + prover
cs = R.chain_r1cs(m, prm.q)
dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2); dev.fill_uniform(asg[:2], 0, 7); dev.chain_assignment(asg, m)
t0 = time.time(); w = dev.witness_map(dcs, asg); torch.cuda.synchronize(); print("witness first call (plan build) %.2f s" % (time.time() - t0), flush=True)
ms = timeit(lambda: dev.witness_map(dcs, asg), reps=2, warm=0)
print("witness_map m=%d: %.2f ms (%.2f us/constraint)" % (m, ms, ms * 1e3 / m), flush=True)
pk = dict(s_pows=crs, delta_ts=dev.fill_uniform(dev.enc_empty(m + 1), 1, 4), delta_mid=dev.fill_uniform(dev.enc_empty(m), 1, 5),
          alpha=dev.fill_uniform(dev.enc_empty(), 1, 6), beta=dev.fill_uniform(dev.enc_empty(), 1, 7))
ms = timeit(lambda: dev.groth16_prove(dcs, pk, asg, want_empty=False), reps=2, warm=1)
print("groth16_prove m=%d: %.2f ms -> %.0f constraints/s ; timings %s" % (m, ms, m / ms * 1e3, dev.last_timings()), flush=True)
