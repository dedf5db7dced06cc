#!/usr/bin/env python3
"""One rank's share of BASELINE.json configs[3] on ONE GPU: Rinocchio, 2^18 constraints, ring N = 16384 with 6 primes,
encodings N_enc = 16384, K = 8, eight ranks.

make_plan(8, rank, 6): two limb groups (3 limbs each) x four term shards.  The rank runs the REAL sharded prover
(ringsnark_amd.dist.rinocchio_prove_sharded on the device backend) on its share -- the whole witness map of its three
limbs at M = 2^18 keeping the rows of its term range (rs_witness_map_rows), its quarter of the eleven inner products
against tiled key windows -- with the two collectives stubbed (one rank: the all-reduce of the partial sums and the
all-gather of the limbs have no peers; this measures one rank's COMPUTE and MEMORY, not the transport).

Checks (tests/proof_check.py, the CPU oracle; untimed):
  * witness map: two slots per limb recomputed through rs_witness_map_slots at full length, (a) identities of every vector
    at random points against the Lagrange form of the interpolants + H Z = A B - C + ZK patch, (b) the rank's kept rows
    equal those columns bit for bit;
  * inner products: one (limb, component, prime) slab of the rank's PARTIAL sum of <alpha_s_pows, H> and of <beta_prods, aux>
    recomputed by the oracle from the device's rows and the key window.

split = slots (the default plan of dist.py): the rank maps its QUARTER of the slots in sub-ranges; the re-shard to terms has
no peers here and is stubbed (the rows of the other ranks' slots stay unwritten), so the inner products are timed on
partly undefined data -- their cost is data independent -- and only the witness map is checked.

usage: tools/c4_rank_rehearsal.py [preset=C4] [logm=18] [rank=5] [logw=10] [zk=1] [world=8] [split=replicate]
Prints one JSON object (copied to profiles/ by hand)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from ringsnark_amd import dist as RD
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from ringsnark_amd.device import Device, to_host

preset = sys.argv[1] if len(sys.argv) > 1 else "C4"
logm = int(sys.argv[2]) if len(sys.argv) > 2 else 18
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 5
logw = int(sys.argv[4]) if len(sys.argv) > 4 else 10
zk = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
world = int(sys.argv[6]) if len(sys.argv) > 6 else 8
split = sys.argv[7] if len(sys.argv) > 7 else "replicate"

GiB = float(2**30)
free0, total = torch.cuda.mem_get_info()
low = [free0]


def mem(tag, log):
    torch.cuda.synchronize()
    f, _ = torch.cuda.mem_get_info()
    low[0] = min(low[0], f)
    log[tag + "_used_gib"] = round((total - f) / GiB, 1)


prm = P.preset(preset)
plan = RD.make_plan(world, rank, prm.L)
prm_l = P.RingParams(prm.N, [prm.q[i] for i in plan.limbs], prm.N_enc, prm.Q, name=prm.name)
m, W = 1 << logm, 1 << logw
out = {"what": "one rank's share of configs[3] on one GPU (collectives stubbed)", "witness_split": split, "preset": preset, "constraints": m, "world": world, "rank": rank,
       "limbs": plan.limbs, "term_shard": "%d of %d" % (plan.term_shard, plan.term_shards), "key_window": W, "zk": zk,
       "ring_primes_two_adicity": [P.two_adicity(q) for q in prm_l.q], "hbm_total_gib": round(total / GiB, 1)}
dev = Device(prm_l)
cs = R.chain_r1cs(m, prm_l.q)
dcs = dev.r1cs(cs)
asg = dev.ring_empty(m + 2)
dev.fill_uniform(asg[:2], 0, 7)
dev.chain_assignment(asg, m)
ranges = RD.rinocchio_key_ranges(plan, m, cs.n_aux)
T = {"s_pows": m + 1, "alpha_s_pows": m + 1, "beta_prods": cs.n_aux}
pk = {}
for i, k in enumerate(("s_pows", "alpha_s_pows", "beta_prods")):
    lo, hi = ranges[k]
    pk[k] = RD.TiledKey(dev.fill_uniform(dev.enc_empty(min(W, hi - lo)), 1, 13 + i), lo, hi, T[k])
for i, k in enumerate(("beta_rv_ts", "beta_rw_ts", "beta_ry_ts")):
    pk[k] = dev.fill_uniform(dev.enc_empty(), 1, 16 + i)
ds = [dev.fill_uniform(dev.ring_empty(), 0, 30 + k) for k in range(3)] if zk else [None] * 3
mem("inputs", out)

# one process: the limb group's all-reduce and the all-gather of the limbs have no peers
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 2000), rank=0, world_size=1)
RD.WITNESS_SPLIT = split
if split == "slots":
    class _NoPeers:
        def __init__(self, *a, **k):
            pass

        def start(self):
            return self

        def finish(self):
            pass
    RD._Exchange = _NoPeers
RD._gather_limbs = lambda plan_, piece, n: piece
_real_all_reduce = dist.all_reduce
RD.dist.all_reduce = lambda *a, **k: None
backend = RD.DeviceBackend(dev)

# ---- setup (excluded from the timed region: SURVEY 8(d) "reported separately"): the per-(context, m) plan -- product-tree
# spectra, rev(Z)^-1 on the host -- and the first-call workspace allocations
t0 = time.time()
w0 = dev.witness_map_slots(dcs, asg, 0, 2, *ds, want=("A_mid", "B_mid", "C_mid", "H"))
torch.cuda.synchronize()
out["setup_s"] = round(time.time() - t0, 1)
del w0

steps = []
kept = {}
orig_witness = backend.witness


def witness_spy(*a, **k):
    w = orig_witness(*a, **k)
    kept.update({n: w[n] for n in ("A_mid", "B_mid", "C_mid", "H")})  # the rank's rows, for the check below
    return w


backend.witness = witness_spy
if split == "slots":  # keep the term-sharded vectors the sharded prover assembles
    orig_sw = RD.sharded_witness

    def sw_spy(*a, **k):
        w = orig_sw(*a, **k)
        kept.update({n: w[n] for n in ("A_mid", "B_mid", "C_mid", "H")})
        return w
    RD.sharded_witness = sw_spy
proof = None
for it in range(2):
    kept.clear()  # the previous step's rows (96 GiB at full size) go back to the allocator before the next step asks for its own
    del proof
    dev.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.time()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    proof, empty = RD.rinocchio_prove_sharded(backend, plan, None, dcs, pk, asg, m, cs.n_inputs, cs.n_aux, *ds)
    ev[1].record()
    torch.cuda.synchronize()
    wall = time.time() - t0
    kern = sorted(dev.profile_read(), key=lambda k: -k["total_ms"])
    dev.set_profiling(False)
    wit = sum(k["total_ms"] for k in kern if not (k["name"].startswith("mac_") or k["name"].startswith("plain_") or k["name"].startswith("reduce")))
    steps.append({"wall_ms": round(wall * 1e3, 1), "device_ms": round(ev[0].elapsed_time(ev[1]), 1), "witness_kernels_ms": round(wit, 1),
                  "inner_product_kernels_ms": round(sum(k["total_ms"] for k in kern) - wit, 1),
                  "kernels": [{"name": k["name"], "ms": round(k["total_ms"], 1), "launches": k["launches"]} for k in kern[:12]]})
    mem("step%d" % it, out)
out["steps"] = steps
out["rank_step_ms"] = steps[-1]["wall_ms"]
out["constraints_per_s_if_all_ranks_take_this_long"] = round(m / (steps[-1]["wall_ms"] * 1e-3))
out["peak_used_gib"] = round((total - low[0]) / GiB, 1)

# ---- checks
from tests import helpers as H
from tests import proof_check

t0 = time.time()
rng = np.random.RandomState(11)
octx = H.oracle_ctx(prm_l)
s0 = 2 * int(rng.randint(prm_l.N // 2 - 1))
if split == "slots":  # a slot pair this rank mapped itself
    b0, bn = plan.slot_range(prm_l.N)
    s0 = b0 + 2 * int(rng.randint(bn // 2 - 1))
wc = dev.witness_map_slots(dcs, asg, s0, 2, *ds, want=("A_mid", "B_mid", "C_mid", "H"))
torch.cuda.synchronize()
errs = []
for limb in range(prm_l.L):
    for slot in (0, 1):
        x = [int(v) for v in to_host(asg[:, limb, s0 + slot].contiguous())]
        g = {k: to_host(wc[k][:, limb, slot].contiguous()) for k in ("A_mid", "B_mid", "C_mid", "H")}
        dv = tuple(0 if d is None else int(to_host(d[limb, s0 + slot].reshape(1))[0]) for d in ds)
        e = proof_check._column_identities(int(prm_l.q[limb]), m, cs, limb, x, g, dv, rng, points=1)
        if e:
            errs.append("%s at limb %d slot %d" % (e, limb, s0 + slot))
lo, hi = plan.term_range(m + 1)
for k in ("A_mid", "B_mid", "C_mid", "H"):
    hi_k = min(hi, m) if k != "H" else hi
    lo_k = min(lo, m) if k != "H" else lo
    if not bool((kept[k][:, :, s0:s0 + 2] == wc[k][lo_k:hi_k]).all()):
        errs.append("kept rows of %s differ from the full-length columns" % k)
out["check_columns_s"] = round(time.time() - t0, 1)
t0 = time.time()
slabs = []
for idx, kname, vec, Tn in (() if split == "slots" else ((7, "alpha_s_pows", kept["H"], hi - lo),) + (((8, "beta_prods", asg[cs.n_inputs:][ranges["beta_prods"][0]:ranges["beta_prods"][1]],
                                                                          ranges["beta_prods"][1] - ranges["beta_prods"][0]),) if not zk else ())):
    l, c, j = int(rng.randint(prm_l.L)), int(rng.randint(2)), int(rng.randint(prm_l.K))
    acc = np.zeros(prm_l.N_enc, dtype=np.uint64)
    proof_check.slab_inner_product(octx, acc, proof_check.key_slab(pk[kname].store, l, c, j, prm_l.N_enc), vec, l, j, Tn)
    if not (acc == to_host(proof[idx, l, c, j].contiguous())).all():
        errs.append("partial sum of proof element %d, slab (limb %d, component %d, prime %d), differs from the CPU oracle" % (idx, l, c, j))
    slabs.append("elem%d[limb %d][comp %d][prime %d] over %d terms" % (idx, l, c, j, Tn))
out["check_slabs_s"] = round(time.time() - t0, 1)
out["check"] = {"ok": not errs, "errors": errs, "columns": "limbs x slots %d,%d: identities at 1 random point each + kept rows == columns" % (s0, s0 + 1),
                "slabs": slabs if split != "slots" else "skipped: the rows of the peers' slots do not exist in a one-rank rehearsal"}
print(json.dumps(out))
dist.destroy_process_group()
sys.exit(0 if not errs else 1)
