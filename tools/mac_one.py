"""One inner product <key, vec> at the headline shape (for rocprofv3 --pmc runs): m terms, key window 2^logw."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ringsnark_amd import params as P, _lib  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

terms = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
logw = int(sys.argv[2]) if len(sys.argv) > 2 else 12
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prm = P.preset("C3")
dev = Device(prm)
for kv in os.environ.get("RS_TUNING", "").split(","):
    if "=" in kv:
        _lib.check(_lib.load().rs_set_tuning(kv.split("=")[0].encode(), int(kv.split("=")[1])))
W = 1 << logw
key = dev.fill_uniform(dev.enc_empty(W), 1, 13)
vecs = [dev.fill_uniform(dev.ring_empty(terms), 0, 7 + g) for g in range(groups)]
dev.set_profiling(True)
for _ in range(2):
    out, _ = dev.msm([key], [(v, None, g) for g, v in enumerate(vecs)], groups, crs_len=terms, window=W)
torch.cuda.synchronize()
for k in dev.profile_read():
    print("  %-32s %4d launches %8.2f ms  %7.1f GB/s alg" % (k["name"], k["launches"], k["total_ms"], k["alg_bytes"] / k["total_ms"] / 1e6))
