"""Where the time of mac_kernel_v3<false, 14> goes at the configs[3] shape (preset C4, Rinocchio, m = 2^12, key window 2^9).
Sweeps the term-chunk knob (other knobs: RS_TUNING=key=value,...); run under RINGSNARK_AMD_LIB=<ablated build> (csrc/Makefile `experiments`, -DRS_MAC3_ABLATE=1|2|4:
wrong results, timing only) to split row traffic / ciphertext traffic / transform.
usage: mac16_probe.py [logm] [preset] [logw] [units,units,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ringsnark_amd import params as P, r1cs as R, _lib  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

logm = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prm = P.preset(sys.argv[2] if len(sys.argv) > 2 else "C4")
W = (1 << int(sys.argv[3])) if len(sys.argv) > 3 else 512
units = [int(u) for u in sys.argv[4].split(",")] if len(sys.argv) > 4 else [768]
dev = Device(prm)
m = 1 << logm
lib = _lib.load()
nk = (lambda T: min(T, W) if W else T)
cs = R.chain_r1cs(m, prm.q)
dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
pk = dict(s_pows=dev.fill_uniform(dev.enc_empty(nk(m + 1)), 1, 3), alpha_s_pows=dev.fill_uniform(dev.enc_empty(nk(m + 1)), 1, 4),
          beta_prods=dev.fill_uniform(dev.enc_empty(nk(m)), 1, 5), beta_rv_ts=dev.fill_uniform(dev.enc_empty(), 1, 6),
          beta_rw_ts=dev.fill_uniform(dev.enc_empty(), 1, 7), beta_ry_ts=dev.fill_uniform(dev.enc_empty(), 1, 8))
for kv in os.environ.get("RS_TUNING", "").split(","):  # e.g. RS_TUNING=mac_share_keys=0
    if "=" in kv:
        _lib.check(lib.rs_set_tuning(kv.split("=")[0].encode(), int(kv.split("=")[1])))
dev.set_profiling(True)
for u in units:
    _lib.check(lib.rs_set_tuning(b"mac_chunk_units", u))
    for it in range(2):
        dev.rinocchio_prove(dcs, pk, asg, window=W)
    torch.cuda.synchronize()
    dev.profile_read()
    t0 = time.time()
    dev.rinocchio_prove(dcs, pk, asg, window=W)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = dev.profile_read()
    tm = dev.last_timings()
    print("lib %s units %d: %.1f ms (witness %.1f, msm %.1f)" % (os.path.basename(_lib.LIB_PATH), u, dt * 1e3, tm["witness_ms"], tm["msm_ms"]), flush=True)
    for k in st[:5]:
        print("    %-40s %5d launches %9.2f ms" % (k["name"], k["launches"], k["total_ms"]), flush=True)
