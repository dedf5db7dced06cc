"""The sharded provers (ringsnark_amd/dist.py) at CONFIGURATION shape: two ranks on one MI355X over gloo, the real
device backend, BASELINE.json configs[2] (ringGroth16, 2^16 constraints, N = 8192, tiled key) and the configs[3]
shape (Rinocchio, N = 16384, 6 ring primes, K = 8).  What this exercises that the toy-scale tests of test_dist.py do
not: TiledKey / key_slice window arithmetic at N = 8192 / 16384, element offsets beyond 2 GiB, the fused per-limb
prover on a limb subset, the term split of a 2^16-term inner product, the all-reduce of real partial sums.

Check: the sharded proof equals, bit for bit, the proof the one-GPU prover computes for the same statement and key, and
THAT proof is checked against the CPU oracle exactly as bench.py checks its timed proof (tests/proof_check.py:
witness-map identities on sampled columns + full proof slabs recomputed by the oracle).

Unmeasured on hardware: the RCCL transport itself (one GPU here, gloo moves the collectives)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ringsnark_amd import dist as RD
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_inputs(dev, prover, m, W, zk):
    """Key (stored on min(T, W) elements per vector: a tiled key when W < T), assignment and blinding on the FULL ring."""
    asg = dev.ring_empty(m + 2)
    dev.fill_uniform(asg[:2], 0, 7)
    dev.chain_assignment(asg, m)
    names = ("s_pows", "delta_ts", "delta_mid") if prover == "groth16" else ("s_pows", "alpha_s_pows", "beta_prods")
    T = {"s_pows": m + 1, "delta_ts": m + 1, "alpha_s_pows": m + 1, "delta_mid": m, "beta_prods": m}
    pk = {k: dev.fill_uniform(dev.enc_empty(min(T[k], W)), 1, 13 + i) for i, k in enumerate(names)}
    singles = ("alpha", "beta") if prover == "groth16" else ("beta_rv_ts", "beta_rw_ts", "beta_ry_ts")
    for i, k in enumerate(singles):
        pk[k] = dev.fill_uniform(dev.enc_empty(), 1, 16 + i)
    ds = [dev.fill_uniform(dev.ring_empty(), 0, 30 + k) for k in range(3)] if zk else [None] * 3
    return asg, pk, ds, T


def _worker(rank, world, port, preset, n_limbs, prover, logm, logw, zk, split, tmp):
    """Each rank derives its share from the SAME full-ring data (generated on a full context, sliced, freed); rank 0
    also computes and oracle-checks the one-GPU proof before the sharded run."""
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    RD.WITNESS_SPLIT = split
    try:
        from ringsnark_amd.device import Device, to_host
        from tests.proof_check import groth16_check, rinocchio_check
        prm = P.preset(preset)
        if n_limbs:
            prm = P.RingParams(prm.N, prm.q[:n_limbs], prm.N_enc, prm.Q, name=prm.name)
        m, W = 1 << logm, 1 << logw
        cs = R.chain_r1cs(m, prm.q)
        plan = RD.make_plan(world, rank, prm.L)
        ranges = (RD.groth16_key_ranges if prover == "groth16" else RD.rinocchio_key_ranges)(plan, m, cs.n_aux)
        single, info, ok = None, None, True
        local = {}
        for turn in range(world):  # one rank at a time holds the full-ring data
            if turn == rank:
                dev = Device(prm, 0)
                asg, pk, ds, T = _full_inputs(dev, prover, m, W, zk)
                window = W if any(min(T[k], W) < T[k] for k in T if k in pk) else 0
                if rank == 0:
                    dcs = dev.r1cs(cs)
                    if prover == "groth16":
                        single = to_host(dev.groth16_prove(dcs, pk, asg, want_empty=False, window=window)[0])
                    else:
                        p9, e9 = dev.rinocchio_prove(dcs, pk, asg, *ds, window=window)
                        single = (to_host(p9), e9)
                # this rank's share: its limbs, and per key vector the stored elements its term range maps to
                for k, v in pk.items():
                    if v.dim() == 5:
                        # this rank reads logical terms [lo, hi); the full-ring prover reads term t from stored[t % len];
                        # the rank stores min(hi - lo, len) of them, arranged so that its term lo + k is store[k % n]
                        lo, hi = ranges[k]
                        n = max(1, min(hi - lo, v.shape[0]))
                        idx = (lo + torch.arange(n, device=v.device)) % v.shape[0]
                        local[k] = RD.TiledKey(v[idx][:, plan.limbs].contiguous(), lo, hi, T[k])
                    else:
                        local[k] = v[plan.limbs].contiguous()
                local["asg"] = asg[:, plan.limbs].contiguous()
                local["ds"] = [None if d is None else d[plan.limbs].contiguous() for d in ds]
                if rank == 0:  # the one-GPU proof against the CPU oracle, exactly as bench.py checks its timed proof
                    if prover == "groth16":
                        ok, info = groth16_check(dev, prm, cs, dcs, asg, dict(pk), torch.from_numpy(single.view(np.int64)).to(dev.device), m,
                                                 window or None, n_cols=2)
                    else:
                        ok, info = rinocchio_check(dev, prm, cs, dcs, asg, pk, torch.from_numpy(single[0].view(np.int64)).to(dev.device), m, ds)
                    del dcs
                del asg, pk, ds, dev
                torch.cuda.empty_cache()
            dist.barrier()
        prm_local = P.RingParams(prm.N, [prm.q[i] for i in plan.limbs], prm.N_enc, prm.Q, name=prm.name)
        dev = Device(prm_local, 0)
        dcs = dev.r1cs(R.chain_r1cs(m, prm_local.q))
        tg = RD.groups_for(plan)
        pk_local = {k: v for k, v in local.items() if k not in ("asg", "ds")}
        if prover == "groth16":
            got = to_host(RD.groth16_prove_sharded(RD.DeviceBackend(dev), plan, tg, dcs, pk_local, local["asg"], m, cs.n_inputs, cs.n_aux))
            same = rank != 0 or bool((got == single).all())
        else:
            got, empty = RD.rinocchio_prove_sharded(RD.DeviceBackend(dev), plan, tg, dcs, pk_local, local["asg"], m, cs.n_inputs, cs.n_aux,
                                                    *local["ds"])
            same = rank != 0 or (bool((to_host(got) == single[0]).all()) and list(empty) == list(single[1]))
        if rank == 0:
            open(tmp, "w").write("ok" if (ok and same) else "oracle check: %s %s; sharded == one-GPU proof: %s" % (ok, info, same))
    finally:
        dist.destroy_process_group()


def _run(tmp_path, *args):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port()) + args + (out,), nprocs=2, join=True)
    assert open(out).read() == "ok", open(out).read()


@pytest.mark.gpu
@pytest.mark.parametrize("n_limbs,split,logm,logw,desc", [
    (None, "replicate", 16, 12, "limb split: 2 of the 4 ring limbs per rank, each rank runs the fused device prover on its limbs"),
    (1, "replicate", 16, 15, "one limb on two ranks: whole witness map on both, 2^15-term halves of every inner product, all-reduce"),
    (1, "slots", 14, 13, "one limb on two ranks: slot-split witness map, row exchange, term-split inner products, all-reduce"),
])
def test_groth16_sharded_at_the_headline_shape(tmp_path, n_limbs, split, logm, logw, desc):
    """BASELINE.json configs[2] (ringGroth16, N = 8192, K = 4; 2^16 constraints for the first two plans) on two ranks."""
    _run(tmp_path, "C3", n_limbs, "groth16", logm, logw, False, split)


@pytest.mark.gpu
@pytest.mark.parametrize("n_limbs,split,zk,logm,desc", [
    (None, "replicate", True, 10, "limb split: 3 of the 6 ring limbs per rank"),
    (1, "replicate", True, 11, "one limb on two ranks: term split of the ten inner products, all-reduce, ZK shifts on the sums"),
    (1, "slots", False, 10, "one limb on two ranks: slot-split witness map and row exchange"),
])
def test_rinocchio_sharded_at_the_c4_shape(tmp_path, n_limbs, split, zk, logm, desc):
    """BASELINE.json configs[3] shape (Rinocchio, N = 16384, 6 ring primes, N_enc = 16384, K = 8) on two ranks, whole
    (untiled) key.  The configuration's 2^18 constraints need a 9 TiB key: the shape is what is exercised."""
    _run(tmp_path, "C4", n_limbs, "rinocchio", logm, 30, zk, split)
