import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ringsnark_amd import _lib, params as P, r1cs as R
from ringsnark_amd.device import Device
prm = P.preset("C3"); dev = Device(prm); m = 65536
cs = R.chain_r1cs(m, prm.q)
asg = dev.ring_empty(m + 2); dev.fill_uniform(asg[:2], 0, 9); dev.chain_assignment(asg, m)
dcs = dev.r1cs(cs)
for _ in range(2): w = dev.witness_map(dcs, asg, None, None, None); del w
torch.cuda.synchronize()
dev.set_profiling(True); dev.profile_read()
w = dev.witness_map(dcs, asg, None, None, None)
torch.cuda.synchronize()
for k in dev.profile_read():
    print("%-34s n=%3d %8.2f ms  %8.3f ms/launch  %8.1f GB/launch-bytes(MB) %8.0f GB/s  %6.2f Tflop" % (k["name"], k["launches"], k["total_ms"], k["total_ms"]/k["launches"], k["alg_bytes"]/k["launches"]/1e6, k["alg_bytes"]/k["total_ms"]/1e6, k["fp64_ops"]/k["total_ms"]/1e9))
