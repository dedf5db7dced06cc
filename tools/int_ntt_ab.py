import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import _lib, params as P
from ringsnark_amd.device import Device
prm = P.preset("micro60"); dev = Device(prm); lib = _lib.load()
batch = (1 << 30) // (prm.N_enc * 8)
src = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, int(prm.Q[0]))
ref = None
for v in (0, 1, 0, 1):
    _lib.check(lib.rs_set_tuning(b"int_ntt_variant", v))
    d = src.clone(); dev.ntt(d, _lib.RS_MOD_COEFF, 0); f = d.clone(); dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=True)
    rt = bool((d == src).all()); ref = f if ref is None else ref
    res = []
    for inv in (False, True):
        for _ in range(5): dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=inv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=inv)
        e1.record(); torch.cuda.synchronize()
        res.append(batch * prm.N_enc * 16 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    print("int_ntt_variant %d: fwd %.0f GB/s inv %.0f GB/s same_as_generic=%s roundtrip=%s" % (v, res[0], res[1], bool((f == ref).all()), rt), flush=True)
