"""Quick probe of the 2^16-constraint headline on one GPU (tiled synthetic key): time per proof and phases."""
import sys
import time

import torch

sys.path.insert(0, ".")
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

logm = int(sys.argv[1]) if len(sys.argv) > 1 else 16
logw = int(sys.argv[2]) if len(sys.argv) > 2 else 13
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
prm = P.preset(sys.argv[4] if len(sys.argv) > 4 else "C3")
dev = Device(prm)
import os  # noqa: E402
from ringsnark_amd import _lib  # noqa: E402
for kv in os.environ.get("RS_TUNING", "").split(","):  # e.g. RS_TUNING=witness_sub_ct=2,ntt_variant=12
    if "=" in kv:
        _lib.check(_lib.load().rs_set_tuning(kv.split("=")[0].encode(), int(kv.split("=")[1])))
m, W = 1 << logm, 1 << logw
t0 = time.time()
cs = R.chain_r1cs(m, prm.q)
dcs = dev.r1cs(cs)
print("r1cs %.1fs" % (time.time() - t0), flush=True)
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
pk = {k: dev.fill_uniform(dev.enc_empty(W), 1, 13 + i) for i, k in enumerate(("s_pows", "delta_ts", "delta_mid"))}
pk["alpha"] = dev.fill_uniform(dev.enc_empty(), 1, 16)
pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 17)
torch.cuda.synchronize()
t0 = time.time()
dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W)
torch.cuda.synchronize()
print("first proof (plans, io cache) %.1fs" % (time.time() - t0), flush=True)
dev.set_profiling(True)
for _ in range(steps):
    t0 = time.time()
    dev.groth16_prove(dcs, pk, asg, want_empty=False, window=W)
    torch.cuda.synchronize()
    print("proof %.1f ms" % ((time.time() - t0) * 1e3), dev.last_timings(), flush=True)
for k in dev.profile_read():
    if k["total_ms"] > 2.0 * steps:
        print("  %-44s %4d launches %8.2f ms/proof" % (k["name"], k["launches"] // steps, k["total_ms"] / steps))
print("peak torch GiB %.1f" % (torch.cuda.max_memory_allocated() / 2**30), "free/total GiB",
      [x / 2**30 for x in torch.cuda.mem_get_info()], flush=True)
