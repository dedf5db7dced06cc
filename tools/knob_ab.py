"""A/B of one rs_set_tuning knob on one GPU: the headline ringGroth16 proof (C3, 2^16 constraints, window 2^13) and the
configs[3]-shape Rinocchio proof (C4, 2^12 constraints, window 2^9), per-kernel times of the inner products.
usage: tools/knob_ab.py <knob> <value,value,...> [groth16|rinocchio|both]      KNOB_AB_KERNELS=prefix,prefix: the kernels listed"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ringsnark_amd import _lib, params as P, r1cs as R
from ringsnark_amd.device import Device

knob, values = sys.argv[1].encode(), [int(v) for v in sys.argv[2].split(",")]
which = sys.argv[3] if len(sys.argv) > 3 else "both"
PREFIXES = tuple(os.environ.get("KNOB_AB_KERNELS", "mac_,plain_").split(","))
lib = _lib.load()
for kv in filter(None, os.environ.get("KNOB_AB_SET", "").split(",")):  # other knobs held fixed: KNOB_AB_SET=key=value,key=value
    _lib.check(lib.rs_set_tuning(kv.split("=")[0].encode(), int(kv.split("=")[1])))


def run(tag, dev, prove, m):
    for v in values:
        _lib.check(lib.rs_set_tuning(knob, v))
        for _ in range(2):
            prove()
        torch.cuda.synchronize()
        dev.set_profiling(True)
        dev.profile_read()
        t0 = time.time()
        for _ in range(3):
            prove()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        ks = dev.profile_read()
        dev.set_profiling(False)
        print("%s %s=%d: %.1f ms/proof (%.0f constraints/s)  " % (tag, knob.decode(), v, dt * 1e3, m / dt) +
              "  ".join("%s %.1f" % (k["name"][:28], k["total_ms"] / 3) for k in ks if k["name"].startswith(PREFIXES)), flush=True)


if which in ("groth16", "both"):
    prm = P.preset(os.environ.get("KNOB_AB_PRESET", "C3"))  # C3 = the recipe primes (the headline); C3F: ring primes = 1 mod 2^20
    dev = Device(prm)
    m, W = 1 << 16, 1 << 13
    dcs = dev.r1cs(R.chain_r1cs(m, prm.q))
    asg = dev.ring_empty(m + 2)
    dev.fill_uniform(asg[:2], 0, 7)
    dev.chain_assignment(asg, m)
    pk = {k: dev.fill_uniform(dev.enc_empty(W), 1, 13 + i) for i, k in enumerate(("s_pows", "delta_ts", "delta_mid"))}
    pk["alpha"], pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 16), dev.fill_uniform(dev.enc_empty(), 1, 17)
    run(prm.name + " groth16 2^16", dev, lambda: dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W), m)
    del dev, dcs, asg, pk
    torch.cuda.empty_cache()
if which in ("rinocchio", "both"):
    prm = P.preset("C4")
    dev = Device(prm)
    m, W = 1 << 12, 1 << 9
    dcs = dev.r1cs(R.chain_r1cs(m, prm.q))
    asg = dev.ring_empty(m + 2)
    dev.fill_uniform(asg[:2], 0, 7)
    dev.chain_assignment(asg, m)
    pk = dict(s_pows=dev.fill_uniform(dev.enc_empty(W), 1, 3), alpha_s_pows=dev.fill_uniform(dev.enc_empty(W), 1, 4),
              beta_prods=dev.fill_uniform(dev.enc_empty(W), 1, 5), beta_rv_ts=dev.fill_uniform(dev.enc_empty(), 1, 6),
              beta_rw_ts=dev.fill_uniform(dev.enc_empty(), 1, 7), beta_ry_ts=dev.fill_uniform(dev.enc_empty(), 1, 8))
    ds = [dev.fill_uniform(dev.ring_empty(), 0, 30 + k) for k in range(3)]
    run("C4 rinocchio 2^12", dev, lambda: dev.rinocchio_prove(dcs, pk, asg, *ds, window=W), m)
