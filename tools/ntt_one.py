import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, _lib
from ringsnark_amd.device import Device
v = int(sys.argv[1]); inv = len(sys.argv) > 2 and sys.argv[2] == "inv"
prm = P.preset("C3"); dev = Device(prm); lib = _lib.load()
_lib.check(lib.rs_set_tuning(b"ntt_variant", v))
d = torch.empty((8192, prm.N_enc), dtype=torch.int64, device=dev.device); d.random_(0, prm.Q[0])
for _ in range(3): dev.ntt(d, _lib.RS_MOD_COEFF, 0, inverse=inv)
torch.cuda.synchronize()
