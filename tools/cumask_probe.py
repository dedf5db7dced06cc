"""Does a CU-masked stream (hipExtStreamCreateWithCUMask) let an HBM-bound kernel keep its bandwidth on half of the CUs
while a VALU-bound kernel runs on the other half?  NTT (HBM-bound) and interpolate at n = 2^13 (VALU-bound) on full and
half-machine streams, alone and concurrently."""
import ctypes as C
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from ringsnark_amd import _lib, params as P  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

prm = P.preset("C3")
dev = Device(prm)
dev2 = Device(prm)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


full = (1 << 256) - 1
# CU numbering of the mask: try "every other CU" (spreads over the XCDs) and "lower half"
patterns = {"even CUs": sum(1 << i for i in range(0, 256, 2)), "odd CUs": sum(1 << i for i in range(1, 256, 2)),
            "low half": (1 << 128) - 1, "high half": ((1 << 128) - 1) << 128}
streams = {k: masked_stream(v) for k, v in patterns.items()}
streams["all"] = masked_stream(full)
batch = (2 << 30) // (prm.N_enc * 8)
polys = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, int(prm.Q[0]))
n = 8192
y = dev2.fill_uniform(dev2.ring_empty(n), 0, 3)
out = torch.empty_like(y)
lib = _lib.load()


def ntt(s, reps):
    for _ in range(reps):
        _lib.check(lib.rs_ntt_forward(dev.h, _lib.RS_MOD_COEFF, 0, C.c_void_p(polys.data_ptr()), batch, s))


def interp(s, reps):
    for _ in range(reps):
        _lib.check(lib.rs_interpolate(dev2.h, C.c_void_p(y.data_ptr()), C.c_void_p(out.data_ptr()), n, s))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


ntt(streams["all"], 5)
interp(streams["all"], 2)
for name in ("all", "even CUs", "low half"):
    print("%-9s NTT x20: %.1f ms   interpolate x4: %.1f ms" % (name, timed(lambda: ntt(streams[name], 20)), timed(lambda: interp(streams[name], 4))), flush=True)
for a, b in (("all", "all"), ("even CUs", "odd CUs"), ("low half", "high half")):
    def both():
        t1 = threading.Thread(target=ntt, args=(streams[a], 20))
        t2 = threading.Thread(target=interp, args=(streams[b], 4))
        t1.start(); t2.start(); t1.join(); t2.join()
    print("concurrent NTT on [%s] + interpolate on [%s]: %.1f ms" % (a, b, timed(both)), flush=True)
