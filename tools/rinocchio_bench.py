"""Rinocchio prover time at m = 2^logm (synthetic key, optionally a window of 2^logw elements), for MAC variants.
usage: rinocchio_bench.py [logm] [preset] [logw]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, r1cs as R, _lib
from ringsnark_amd.device import Device
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prm = P.preset(sys.argv[2] if len(sys.argv) > 2 else "C3"); dev = Device(prm); m = 1 << logm; lib = _lib.load()
W = (1 << int(sys.argv[3])) if len(sys.argv) > 3 else 0
nk = (lambda T: min(T, W) if W else T)
cs = R.chain_r1cs(m, prm.q); dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2); dev.fill_uniform(asg[:2], 0, 7); dev.chain_assignment(asg, m)
pk = dict(s_pows=dev.fill_uniform(dev.enc_empty(nk(m + 1)), 1, 3), alpha_s_pows=dev.fill_uniform(dev.enc_empty(nk(m + 1)), 1, 4),
          beta_prods=dev.fill_uniform(dev.enc_empty(nk(m)), 1, 5), beta_rv_ts=dev.fill_uniform(dev.enc_empty(), 1, 6),
          beta_rw_ts=dev.fill_uniform(dev.enc_empty(), 1, 7), beta_ry_ts=dev.fill_uniform(dev.enc_empty(), 1, 8))
dev.set_profiling(True)
for variant in (2, 3, 5):
    _lib.check(lib.rs_set_tuning(b"mac_variant", variant))
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        dev.rinocchio_prove(dcs, pk, asg, window=W); torch.cuda.synchronize()
        dt = time.time() - t0
    print("%s mac_variant %d: rinocchio prove m=%d: %.1f ms -> %.0f constraints/s %s" % (prm.name, variant, m, dt * 1e3, m / dt, dev.last_timings()), flush=True)
for k in dev.profile_read()[:8]:
    print("  %-40s %5d launches %9.2f ms total  %7.1f GB/s alg" % (k["name"], k["launches"], k["total_ms"], k["alg_bytes"] / max(k["total_ms"], 1e-9) / 1e6))
