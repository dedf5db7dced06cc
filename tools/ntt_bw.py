"""Standalone forward NTT bandwidth on a 1 GiB batch (row a4 roofline): 16 algorithmic bytes per coefficient."""
import sys

import torch

sys.path.insert(0, ".")
from ringsnark_amd import _lib, params as P  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

prm = P.preset(sys.argv[1] if len(sys.argv) > 1 else "C3")
dev = Device(prm)
if len(sys.argv) > 2:
    _lib.check(_lib.load().rs_set_tuning(b"ntt_variant", int(sys.argv[2])))
if len(sys.argv) > 3:
    _lib.check(_lib.load().rs_set_tuning(b"ntt_wide_grid", int(sys.argv[3])))
for gib in (0.25, 0.5, 1, 2, 4, 1):
    batch = int(gib * (1 << 30)) // (prm.N_enc * 8)
    polys = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, int(prm.Q[0]))
    for inverse in (False, True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            dev.ntt(polys, _lib.RS_MOD_COEFF, 0, inverse=inverse)
        e0.record()
        for _ in range(20):
            dev.ntt(polys, _lib.RS_MOD_COEFF, 0, inverse=inverse)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%5.2f GiB %s: %7.3f ms  %6.1f GB/s  frac %.3f  (%.1f ns/transform)" % (
            gib, "inv" if inverse else "fwd", ms, batch * prm.N_enc * 16 / ms / 1e6, batch * prm.N_enc * 16 / ms / 1e6 / 8000, ms * 1e6 / batch))
    del polys
