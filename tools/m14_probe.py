"""Witness map at 2^13 < m <= 2^14 on the recipe primes (preset C3): multi-pass path with incomplete transforms (witness_inc = 1)
against the pairwise block convolutions these sizes took before (witness_inc = 0).  usage: gpurun -- python tools/m14_probe.py"""
import sys, time
sys.path.insert(0, ".")
import torch
from ringsnark_amd import params as P, r1cs as R, _lib
from ringsnark_amd.device import Device
lib=_lib.load()
for inc in (1, 0):
    _lib.check(lib.rs_set_tuning(b"witness_inc", inc))
    prm = P.preset("C3"); dev = Device(prm)
    for m in (12000, 16384):
        cs = R.chain_r1cs(m, prm.q); dcs = dev.r1cs(cs)
        asg = dev.ring_empty(m + 2); dev.fill_uniform(asg[:2], 0, 7); dev.chain_assignment(asg, m)
        for _ in range(2): dev.witness_map(dcs, asg, want=("A_mid","B_mid","H"))
        torch.cuda.synchronize(); t0=time.time()
        for _ in range(3): dev.witness_map(dcs, asg, want=("A_mid","B_mid","H"))
        torch.cuda.synchronize(); print("witness_inc=%d m=%d: %.1f ms per witness map" % (inc, m, (time.time()-t0)/3*1e3), flush=True)
    del dev
