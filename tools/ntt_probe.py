"""Standalone NTT kernel shapes on one GPU (rs_set_tuning "ntt_variant"): one tool for what used to be four scripts.

  ntt_probe.py variants [preset] [v,v,...]   every variant: forward equals variant 0, round trip, forward / inverse GB/s
  ntt_probe.py one <variant> [inv] [preset]  three launches of one variant (the command to put under rocprofv3)
  ntt_probe.py repeat [v,v,...]              compute-only cost: 1 / 5 / 9 in-LDS repetitions ("ntt_repeat"; experiments build)
  ntt_probe.py stagger [v,v,...]             the same knob's stagger field (bits 8+): workgroups start 0..16 transforms apart
(bandwidth sweeps over batch sizes: tools/ntt_bw.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ringsnark_amd import _lib, params as P
from ringsnark_amd.device import Device


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


mode = sys.argv[1] if len(sys.argv) > 1 else "variants"
lib = _lib.load()
tune = lambda k, v: _lib.check(lib.rs_set_tuning(k, v))
ints = lambda s: [int(x) for x in s.split(",")]
B = 8192
if mode == "one":
    v, inv = int(sys.argv[2]), len(sys.argv) > 3 and sys.argv[3] == "inv"
    prm = P.preset(sys.argv[4] if len(sys.argv) > 4 else "C3")
    dev = Device(prm)
    tune(b"ntt_variant", v)
    d = torch.empty((B, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, prm.Q[0])
    for _ in range(3):
        dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=inv)
    torch.cuda.synchronize()
elif mode == "variants":
    prm = P.preset(sys.argv[2] if len(sys.argv) > 2 else "C3")
    dev = Device(prm)
    src = torch.empty((B, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, prm.Q[0])
    ref_f = None
    gb = B * prm.N_enc * 16 / 1e9
    for v in (ints(sys.argv[3]) if len(sys.argv) > 3 else range(15)):
        try:
            tune(b"ntt_variant", v)
        except _lib.RsError as e:
            print("variant %d: %s" % (v, e))
            continue
        d = src.clone()
        dev.ntt(d, _lib.RS_MOD_COEFF, 0)
        f = d.clone()
        dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=True)
        ok_rt = bool((d == src).all())
        ref_f = f if ref_f is None else ref_f
        w = src.clone()
        msf = timeit(lambda: dev.ntt(w, _lib.RS_MOD_COEFF, 0))
        msi = timeit(lambda: dev.ntt(w, _lib.RS_MOD_COEFF, 0, inverse=True))
        print("variant %d: fwd %.3f ms %.0f GB/s | inv %.3f ms %.0f GB/s | same_as_first=%s roundtrip=%s" % (
            v, msf, gb / msf * 1e3, msi, gb / msi * 1e3, bool((f == ref_f).all()), ok_rt), flush=True)
elif mode in ("repeat", "stagger"):
    prm = P.preset("C3")
    dev = Device(prm)
    d = torch.empty((B, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, prm.Q[0])
    for v in (ints(sys.argv[2]) if len(sys.argv) > 2 else (8, 9, 13)):
        tune(b"ntt_variant", v)
        if mode == "repeat":
            res = []
            for rep in (1, 5, 9):
                tune(b"ntt_repeat", rep)
                res.append(timeit(lambda: dev.ntt(d, _lib.RS_MOD_COEFF, 0)))
            print("variant %d: 1x %.3f ms (%.0f GB/s), 5x %.3f, 9x %.3f -> %.1f ns per extra in-LDS transform" % (
                v, res[0], B * 8192 * 16 / 1e6 / res[0], res[1], res[2], (res[2] - res[0]) / 8 / B * 1e6), flush=True)
        else:
            for stag in (0, 1, 2, 4, 8, 16):
                tune(b"ntt_repeat", 1 + (stag << 8))
                ms = timeit(lambda: dev.ntt(d, _lib.RS_MOD_COEFF, 0))
                print("variant %d stagger %d: %.3f ms  %.0f GB/s" % (v, stag, ms, B * 8192 * 16 / 1e6 / ms), flush=True)
    tune(b"ntt_repeat", 1)
else:
    sys.exit(__doc__)
