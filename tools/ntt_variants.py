"""Compare NTT kernel shapes (rs_set_tuning ntt_variant) on one GPU: correctness vs variant 0 + GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, _lib
from ringsnark_amd.device import Device

def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
prm = P.preset(name); dev = Device(prm); lib = _lib.load()
batch = 8192
src = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device); src.random_(0, prm.Q[0])
ref_f = ref_i = None
for v in [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else range(13))]:
    _lib.check(lib.rs_set_tuning(b"ntt_variant", v))
    d = src.clone(); dev.ntt(d, _lib.RS_MOD_COEFF, 0); f = d.clone()
    dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=True)
    ok_rt = bool((d == src).all())
    if ref_f is None: ref_f = f
    ok_f = bool((f == ref_f).all())
    w = src.clone()
    msf = timeit(lambda: dev.ntt(w, _lib.RS_MOD_COEFF, 0))
    msi = timeit(lambda: dev.ntt(w, _lib.RS_MOD_COEFF, 0, inverse=True))
    gb = batch * prm.N_enc * 16 / 1e9
    print("variant %d: fwd %.3f ms %.0f GB/s | inv %.3f ms %.0f GB/s | same_as_v0=%s roundtrip=%s" % (v, msf, gb / msf * 1e3, msi, gb / msi * 1e3, ok_f, ok_rt), flush=True)
