"""Compute-only cost of a transform: time with `repeat` in-LDS repetitions (ntt_repeat knob)."""
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
B = 8192
d = torch.empty((B, prm.N_enc), dtype=torch.int64, device=dev.device); d.random_(0, prm.Q[0])
for v in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else (8, 9, 13))]:
    _lib.check(lib.rs_set_tuning(b"ntt_variant", v))
    res = []
    for rep in (1, 5, 9):
        _lib.check(lib.rs_set_tuning(b"ntt_repeat", rep))
        res.append(timeit(lambda: dev.ntt(d, _lib.RS_MOD_COEFF, 0)))
    per = (res[2] - res[0]) / 8 / B * 1e6
    print("variant %d: 1x %.3f ms (%.0f GB/s), 5x %.3f, 9x %.3f -> %.1f ns per extra in-LDS transform" % (v, res[0], B * 8192 * 16 / 1e6 / res[0], res[1], res[2], per), flush=True)
_lib.check(lib.rs_set_tuning(b"ntt_repeat", 1))
