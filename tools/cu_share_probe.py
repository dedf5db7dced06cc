"""How do the prover's two kinds of kernel scale with the share of the CUs they get?  (round-4 verdict, item 3: overlapping
the HBM-bound passes with the FP64-bound tile kernels can only pay if a streaming pass keeps most of its bandwidth on a
small share of the CUs while the tile kernels lose only that share.)

One headline proof (C3, 2^16 constraints, tiled key) per CU mask -- hipExtStreamCreateWithCUMask, the LOW k/8 of the 256
mask bits (an interleaved pattern, "every other bit", changes nothing on this driver: tools/cumask_probe.py, and the first
run of this probe, profiles/r05_cu_share_probe_interleaved.txt) -- with the library's per-kernel profile; printed per
kernel class.  Then the arithmetic: best spatial split  min_f max(T_valu(1 - f), T_hbm(f))  against the serial sum.
usage: gpurun -- python tools/cu_share_probe.py [logm=16]"""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, ".")
from ringsnark_amd import _lib, params as P, r1cs as R  # noqa: E402
from ringsnark_amd.device import Device  # noqa: E402

logm = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prm = P.preset("C3")
dev = Device(prm)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(keep_of_8):
    bits = (1 << (32 * keep_of_8)) - 1
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


m, W = 1 << logm, 1 << 12
cs = R.chain_r1cs(m, prm.q)
dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
pk = {k: dev.fill_uniform(dev.enc_empty(W), 1, 13 + i) for i, k in enumerate(("s_pows", "delta_ts", "delta_mid"))}
pk["alpha"], pk["beta"] = dev.fill_uniform(dev.enc_empty(), 1, 16), dev.fill_uniform(dev.enc_empty(), 1, 17)
torch.cuda.synchronize()
lib = _lib.load()
proof = dev.enc_empty(3)
s_pk = _lib.Groth16PK(pk["s_pows"].data_ptr(), pk["delta_ts"].data_ptr(), pk["delta_mid"].data_ptr(), pk["alpha"].data_ptr(),
                      pk["beta"].data_ptr(), W, 0)


def prove(stream):
    _lib.check(lib.rs_groth16_prove(dev.h, dcs.h, C.byref(s_pk), C.c_void_p(asg.data_ptr()), C.c_void_p(proof.data_ptr()), None, stream))
    _lib.check(lib.rs_sync(dev.h, stream))


CLASSES = (("tree", ("tree_",)), ("sub-transforms", ("sub_ntt",)), ("mac", ("mac_kernel",)), ("plain rows", ("plain_center",)),
           ("cross passes", ("cross_kernel",)), ("io/mid, eval, transposes", ("io_mid", "r1cs_eval", "transpose")))
VALU = ("tree", "sub-transforms", "mac", "plain rows")
out = {}
for keep in (8, 7, 6, 4, 2, 1):
    st = masked_stream(keep)
    prove(st)
    dev.set_profiling(True)
    dev.profile_read()
    prove(st)
    stats = dev.profile_read()
    dev.set_profiling(False)
    row = {}
    for name, pre in CLASSES:
        row[name] = round(sum(k["total_ms"] for k in stats if k["name"].startswith(pre)), 1)
    row["other"] = round(sum(k["total_ms"] for k in stats) - sum(row.values()), 1)
    row["total"] = round(sum(k["total_ms"] for k in stats), 1)
    out["%d/8 of the CUs" % keep] = row
    print("%d/8 CUs: %s" % (keep, json.dumps(row)), flush=True)
full = out["8/8 of the CUs"]
t_valu = {k: sum(v[c] for c in VALU) for k, v in out.items()}
t_hbm = {k: v["total"] - t_valu[k] for k, v in out.items()}
print("FP64-bound classes: %s" % json.dumps({k: round(v, 1) for k, v in t_valu.items()}))
print("HBM-bound classes:  %s" % json.dumps({k: round(v, 1) for k, v in t_hbm.items()}))
serial = full["total"]
for keep in (7, 6, 4):  # tile kernels on keep/8, streaming passes on the rest
    rest = 8 - keep
    a, b = t_valu["%d/8 of the CUs" % keep], t_hbm.get("%d/8 of the CUs" % rest)
    if b is not None:
        a, b = a, t_hbm["%d/8 of the CUs" % rest]
        print("spatial split %d/8 + %d/8: max(%.1f, %.1f) = %.1f ms against %.1f ms serial (%.3f)" % (keep, rest, a, b, max(a, b), serial, max(a, b) / serial))
