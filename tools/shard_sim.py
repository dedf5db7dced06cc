"""Single-GPU rehearsal of one rank of the sharded prover: L_local ring limbs, m constraints."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ringsnark_amd import params as P, r1cs as R
from ringsnark_amd.device import Device
L_local = int(sys.argv[1]); logm = int(sys.argv[2]); tshare = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prm = P.preset("C3"); prm = P.RingParams(prm.N, prm.q[:L_local], prm.N_enc, prm.Q)
dev = Device(prm); m = 1 << logm
if len(sys.argv) > 4:
    from ringsnark_amd import _lib
    _lib.check(_lib.load().rs_set_tuning(b"witness_lds_logM", int(sys.argv[4])))
cs = R.chain_r1cs(m, prm.q); dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2); dev.fill_uniform(asg[:2], 0, 7); dev.chain_assignment(asg, m)
mt = m // tshare
pk = dict(s_pows=dev.fill_uniform(dev.enc_empty(mt + 1), 1, 3), delta_ts=dev.fill_uniform(dev.enc_empty(mt + 1), 1, 4),
          delta_mid=dev.fill_uniform(dev.enc_empty(mt), 1, 5))
torch.cuda.synchronize()
print("L_local", L_local, "m", m, "term share 1/%d" % tshare, "mem GiB", torch.cuda.memory_allocated() / 2**30, flush=True)
for it in range(2):
    t0 = time.time()
    w = dev.witness_map(dcs, asg, want=("A_io", "A_mid", "B_io", "B_mid", "H")); torch.cuda.synchronize()
    t1 = time.time()
    ab, _ = dev.msm([pk["s_pows"][:mt]], [(w["A_io"][:mt], None, 0), (w["A_mid"][:mt], None, 0), (w["B_io"][:mt], None, 1), (w["B_mid"][:mt], None, 1)], 2)
    c, _ = dev.msm([pk["delta_ts"]], [(w["H"][:mt + 1], None, 0)], 1)
    c2, _ = dev.msm([pk["delta_mid"]], [(asg[2:2 + mt], None, 0)], 1)
    torch.cuda.synchronize(); t2 = time.time()
    print("iter %d: witness %.1f ms, msm %.1f ms -> %.0f constraints/s per rank-step" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, m / (t2 - t0)), flush=True)
    del w, ab, c, c2
