"""A few launches of the standalone transform at one length, for counter passes:
   rocprofv3 --kernel-trace --pmc ... -- python3 tools/ntt16_pmc.py [preset=C4] [gib=1] [launches=3]"""
import sys

import torch

sys.path.insert(0, ".")
from ringsnark_amd import _lib, params as P  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

prm = P.preset(sys.argv[1] if len(sys.argv) > 1 else "C4")
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = Device(prm)
batch = int(gib * (1 << 30)) // (prm.N_enc * 8)
polys = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, int(prm.Q[0]))
for inverse in (False, True):
    for _ in range(reps):
        dev.ntt(polys, _lib.RS_MOD_COEFF, 0, inverse=inverse)
torch.cuda.synchronize()
print("done", batch)
