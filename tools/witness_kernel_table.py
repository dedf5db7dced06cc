"""Per-kernel table of ONE headline witness map (C3, 2^16 constraints, all seven vectors): launches, ms, algorithmic
bytes per launch, algorithmic GB/s and model FP64 rate -- what the HBM-bound passes of the witness map reach.
usage: gpurun -- python tools/witness_kernel_table.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ringsnark_amd import params as P, r1cs as R
from ringsnark_amd.device import Device

prm = P.preset("C3")
dev = Device(prm)
m = 65536
cs = R.chain_r1cs(m, prm.q)
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 9)
dev.chain_assignment(asg, m)
dcs = dev.r1cs(cs)
for _ in range(2):
    w = dev.witness_map(dcs, asg, None, None, None)
    del w
torch.cuda.synchronize()
dev.set_profiling(True)
dev.profile_read()
w = dev.witness_map(dcs, asg, None, None, None)
torch.cuda.synchronize()
for k in dev.profile_read():
    print("%-34s n=%3d %8.2f ms  %8.3f ms/launch  %9.1f MB/launch  %6.0f GB/s  %6.2f T FP64 lane-ops/s"
          % (k["name"], k["launches"], k["total_ms"], k["total_ms"] / k["launches"], k["alg_bytes"] / k["launches"] / 1e6,
             k["alg_bytes"] / k["total_ms"] / 1e6, k["fp64_ops"] / k["total_ms"] / 1e9))
