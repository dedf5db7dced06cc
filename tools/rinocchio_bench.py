"""Rinocchio prover time at C3, m = 2^logm (synthetic key), for MAC variants."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, r1cs as R, _lib
from ringsnark_amd.device import Device
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prm = P.preset("C3"); dev = Device(prm); m = 1 << logm; lib = _lib.load()
cs = R.chain_r1cs(m, prm.q); dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2); dev.fill_uniform(asg[:2], 0, 7); dev.chain_assignment(asg, m)
pk = dict(s_pows=dev.fill_uniform(dev.enc_empty(m + 1), 1, 3), alpha_s_pows=dev.fill_uniform(dev.enc_empty(m + 1), 1, 4),
          beta_prods=dev.fill_uniform(dev.enc_empty(m), 1, 5), beta_rv_ts=dev.fill_uniform(dev.enc_empty(), 1, 6),
          beta_rw_ts=dev.fill_uniform(dev.enc_empty(), 1, 7), beta_ry_ts=dev.fill_uniform(dev.enc_empty(), 1, 8))
for variant in (2, 3, 5):
    _lib.check(lib.rs_set_tuning(b"mac_variant", variant))
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        dev.rinocchio_prove(dcs, pk, asg); torch.cuda.synchronize()
        dt = time.time() - t0
    print("mac_variant %d: rinocchio prove m=%d: %.1f ms -> %.0f constraints/s" % (variant, m, dt * 1e3, m / dt), flush=True)
