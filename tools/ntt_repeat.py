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
prm = P.preset("C3"); dev = Device(prm); lib = _lib.load()
d = torch.empty((8192, prm.N_enc), dtype=torch.int64, device=dev.device); d.random_(0, prm.Q[0])
for v in (8, 9):
    _lib.check(lib.rs_set_tuning(b"ntt_variant", v))
    for stag in (0, 1, 2, 4, 8, 16):
        _lib.check(lib.rs_set_tuning(b"ntt_repeat", 1 + (stag << 8)))
        ms = timeit(lambda: dev.ntt(d, _lib.RS_MOD_COEFF, 0))
        print("variant %d stagger %d: %.3f ms  %.0f GB/s" % (v, stag, ms, 8192 * 8192 * 16 / 1e6 / ms), flush=True)
