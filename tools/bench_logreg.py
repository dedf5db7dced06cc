"""BASELINE.json configs[4]: the reference's logistic-regression inference circuit
(benchmarks/bench_logistic_regression_inference.cpp: 256 features, 1031 constraints, 2055 variables, 517 public) proven
with the Rinocchio prover at the file's own parameters (preset C5: ring N = 2048, the 54-bit BFVDefault(2048) prime,
encodings N_enc = 16384, K = 8).  Synthetic key (uniform residues) and synthetic input ciphertexts.
Prints one JSON line; reported separately from bench.py's headline.

  python tools/bench_logreg.py [steps]
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prm = P.preset("C5")
dev = Device(prm)
F = 256
cs = R.logreg_r1cs(prm.q, F)
inputs = dev.fill_uniform(dev.ring_empty(4 * F), 0, 41)
asg = R.logreg_assignment(F, inputs, dev.ring_mul, dev.ring_add, dev.ring_mul_scalar)
pk = {"s_pows": dev.fill_uniform(dev.enc_empty(cs.m + 1), 1, 22), "alpha_s_pows": dev.fill_uniform(dev.enc_empty(cs.m + 1), 1, 23),
      "beta_prods": dev.fill_uniform(dev.enc_empty(cs.n_aux), 1, 24)}
for i, k in enumerate(("beta_rv_ts", "beta_rw_ts", "beta_ry_ts")):
    pk[k] = dev.fill_uniform(dev.enc_empty(), 1, 25 + i)
ds = [dev.fill_uniform(dev.ring_empty(), 0, 30 + k) for k in range(3)]
dcs = dev.r1cs(cs)
for _ in range(2):
    dev.rinocchio_prove(dcs, pk, asg, *ds)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    dev.rinocchio_prove(dcs, pk, asg, *ds)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
dev.set_profiling(True)
dev.profile_read()
dev.rinocchio_prove(dcs, pk, asg, *ds)
torch.cuda.synchronize()
tm = dev.last_timings()
stats = dev.profile_read()
print(json.dumps({
    "metric": "Rinocchio prover, logistic-regression inference circuit (configs[4])", "value": round(cs.m / dt, 1), "unit": "constraints/s",
    "ms_per_proof": round(dt * 1e3, 3), "constraints": cs.m, "variables": cs.n_vars, "public": cs.n_inputs, "preset": "C5",
    "arithmetic": "hybrid: u64 Montgomery on the ring side (54-bit ring prime), exact FP64 on the encoding side (48/49-bit data primes)", "phase_ms": {"witness_map": round(tm["witness_ms"], 3), "msm": round(tm["msm_ms"], 3)},
    "kernels": [{"name": k["name"], "ms": round(k["total_ms"], 3)} for k in stats[:6]], "data": "synthetic"}))
