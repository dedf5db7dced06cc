import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, _lib
from ringsnark_amd.device import Device
prm = P.preset("C3"); dev = Device(prm); lib = _lib.load()
T = 4096
crs = dev.enc_empty(T); dev.fill_uniform(crs, 1, 3)
v = dev.ring_empty(T); dev.fill_uniform(v, 0, 10)
dev.set_profiling(True)
for variant in (1, 2, 3):
    _lib.check(lib.rs_set_tuning(b"mac_variant", variant))
    for ab in ((0,) if variant == 1 else (0, 1, 2, 6)):
        _lib.check(lib.rs_set_tuning(b"mac_ablate", ab))
        for _ in range(2):
            import ctypes
            dev.lib.rs_groth16_prove  # noqa
            t0 = dev.last_timings()
            dev.inner_product(crs, v, want_used=False); torch.cuda.synchronize()
        # timings accumulate msm_mac_ms across calls since the last prover call; measure delta
        a = dev.last_timings()["msm_mac_ms"]
        dev.inner_product(crs, v, want_used=False); torch.cuda.synchronize()
        b = dev.last_timings()["msm_mac_ms"]
        ms = b - a
        print("variant %d ablate %d: mac %.3f ms  (%.1f ns/unit, ct stream %.0f GB/s)" % (variant, ab, ms, ms * 1e6 / (T * 16), T * prm.enc_words * 8 / 1e6 / ms), flush=True)
