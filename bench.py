#!/usr/bin/env python3
"""bench.py -- ringGroth16 prover throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one groth16::prover call (witness map + all encoding inner products) on a synthetic
chain R1CS (x_i * x_{i+1} = x_{i+2}) with the proving key (synthetic CRS: uniform
residues, legitimate because prover cost is data independent) and the assignment already resident
in HBM.  Workload: headline ring shape C3 (N=8192, L=4 ring primes, N_enc=8192, K=4) with
m = 2^13 constraints PER GPU (48 GiB of proving key per GPU).  For N > 1 ONE proof of m = 2^13 * N
constraints is sharded over limbs, then over constraint ranges (ringsnark_amd/dist.py), so the
per-GPU share of the encoding inner products is constant: weak scaling; N = 8 is the BASELINE.json
headline configuration (2^16 constraints, N = 8192, 4 RNS primes).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(prm, m, budget_terms=64, m_s=96, slots_s=256):
    """The CPU oracle (a scalar port of the reference's algorithm, 1 thread) timed on a bounded
    sample of the same workload: `budget_terms` inner-product terms at full ring shape, and the
    reference's O(m^2) witness map at m_s constraints on slots_s slots of limb 0, scaled by
    (m/m_s)^2 * (N*L/slots_s) (its cost is exactly quadratic in m and linear in slots)."""
    import numpy as np

    from oracle import oracle as O
    from tests import helpers as H

    ctx = H.oracle_ctx(prm)
    encs, rings = ctx.random_enc(1, budget_terms), ctx.random_ring(2, budget_terms)
    t0 = time.perf_counter()
    ctx.inner_product(encs, rings)
    t_term = (time.perf_counter() - t0) / budget_terms
    cs = R.chain_r1cs(m_s, prm.q[:1])
    q = prm.q[0]
    rng = np.random.RandomState(3)
    asg = np.zeros((m_s + 2, slots_s), dtype=np.uint64)
    asg[0] = rng.randint(1, 2**31, slots_s)
    asg[1] = rng.randint(1, 2**31, slots_s)
    for i in range(m_s):
        asg[i + 2] = (asg[i].astype(object) * asg[i + 1].astype(object) % q).astype(np.uint64)
    t0 = time.perf_counter()
    O.witness_map(q, H.oracle_cs(cs), 0, asg)
    t_w = (time.perf_counter() - t0) * (m / m_s) ** 2 * (prm.N * prm.L / slots_s)
    terms = 4 * m + (m + 1) + m  # groth16.tcc:89-112 with n_aux = m
    total = t_w + terms * t_term
    return {
        "value": m / total, "unit": "constraints/s", "cores": 1, "kind": "port",
        "sample": "oracle/rs_oracle.c, 1 thread: inner_product on %d terms at full shape (%.1f ms/term x %d terms) + "
                  "O(m^2) witness map measured at m=%d on %d slots and extrapolated x(m/%d)^2 x(N*L/%d) (%.0f s)"
                  % (budget_terms, t_term * 1e3, terms, m_s, slots_s, m_s, slots_s, t_w),
    }


def mac_algorithmic_bytes(prm, m, n_aux):
    """Algorithmic bytes of the three mac_kernel launches of one proof (DESIGN.md "Roofline"):
    every ciphertext word once, every centred plaintext row once, every accumulator set once."""
    enc = prm.enc_words * 8
    crow = prm.L * prm.N_enc * 8
    launches = [(m, 2), (m + 1, 1)] + ([(n_aux, 1)] if n_aux else [])
    return sum(T * (enc + ng * crow) + ng * enc for T, ng in launches), len(launches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preset", default="C3")
    ap.add_argument("--logm", type=int, default=13, help="log2 of the constraints per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    import torch.distributed as dist

    from ringsnark_amd import dist as RD
    from ringsnark_amd.device import Device

    # Developer rehearsal of the N > 1 path on a single-GPU box: RINGSNARK_BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and uses gloo (the numbers mean nothing; the code path is the production one).
    rehearsal = os.environ.get("RINGSNARK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    prm = P.preset(args.preset)
    m = (1 << args.logm) * world
    plan = RD.make_plan(world, rank, prm.L)
    prm_local = P.RingParams(prm.N, [prm.q[i] for i in plan.limbs], prm.N_enc, prm.Q, name=prm.name)
    dev = Device(prm_local, local_rank)
    cs = R.chain_r1cs(m, prm_local.q)
    dcs = dev.r1cs(cs)
    n_aux = cs.n_aux
    # synthetic inputs, generated on device (seeded per (limb group, role))
    seed0 = 1000 * (plan.limb_group + 1)
    asg = dev.ring_empty(m + 2)
    dev.fill_uniform(asg[:2], 0, seed0 + 7)
    dev.chain_assignment(asg, m)
    # Every rank allocates only the slice of the key it reads: its limbs, and (when limbs are shared)
    # a view positioned so that its term range [lo, hi) lands on real storage.
    ranges = RD.groth16_key_ranges(plan, m, n_aux)

    def key_vector(name, T, seed):
        lo, hi = (0, T) if world == 1 else ranges[name]
        store = dev.fill_uniform(dev.enc_empty(max(hi - lo, 1)), 1, seed + 100 * plan.term_shard)
        return RD.TermWindow(store, lo, hi, T)

    pk = {
        "s_pows": key_vector("s_pows", m + 1, seed0 + 13),
        "delta_ts": key_vector("delta_ts", m + 1, seed0 + 14),
        "delta_mid": key_vector("delta_mid", n_aux, seed0 + 15),
        "alpha": dev.fill_uniform(dev.enc_empty(), 1, seed0 + 16),
        "beta": dev.fill_uniform(dev.enc_empty(), 1, seed0 + 17),
    }
    term_group = RD.groups_for(plan) if world > 1 else None
    backend = RD.DeviceBackend(dev)

    pk1 = {k: (v.store if isinstance(v, RD.TermWindow) else v) for k, v in pk.items()}

    def step():
        if world == 1:
            return dev.groth16_prove(dcs, pk1, asg, want_empty=False)[0]
        return RD.groth16_prove_sharded(backend, plan, term_group, dcs, pk, asg, m, cs.n_inputs, n_aux)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = m * args.steps / elapsed

    # dominant kernel (mac_kernel) timed live with HIP events on the launch stream
    roofline = None
    timings = None
    if world == 1:
        dev.set_profiling(True)
        step()
        torch.cuda.synchronize()
        timings = dev.last_timings()
        dev.set_profiling(False)
        nbytes, nl = mac_algorithmic_bytes(prm, m, n_aux)
        if timings["msm_mac_ms"] > 0:
            achieved = nbytes / (timings["msm_mac_ms"] * 1e-3) / 1e9
            nlaunch = max(1, timings["msm_mac_launches"])
            # HBM traffic of the same kernel from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB units;
            # tools/pmc_summary.py), collected offline with rocprofv3 --pmc on this exact configuration
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s_m%d.json" % (prm.name, m))
            if os.path.exists(pmc):
                ks = [v for n, v in json.load(open(pmc))["kernels"].items() if n.startswith("rs::mac_kernel_v2")]
                k = ks[0] if ks else None
                if k:
                    traffic = int(k["hbm_bytes"] / k["launches_per_proof"])
            roofline = {"bound": "hbm", "kernel": "mac_kernel_v2", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "launches": nlaunch, "avg_launch_ms": round(timings["msm_mac_ms"] / nlaunch, 3),
                        "algorithmic_bytes_per_launch": nbytes // nlaunch}

    # the standalone transform (row a4), HBM bound by design: 16 bytes per coefficient per transform
    ntt_roofline = None
    if world == 1:
        from ringsnark_amd import _lib
        batch = 4096
        polys = torch.empty((batch, prm.N_enc), dtype=torch.int64, device=dev.device).random_(0, int(prm.Q[0]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            dev.ntt(polys, _lib.RS_MOD_COEFF, 0)
        reps = 10
        e0.record()  # the library launches on torch's current stream (device.py passes it down)
        for _ in range(reps):
            dev.ntt(polys, _lib.RS_MOD_COEFF, 0)
        e1.record()
        torch.cuda.synchronize()
        gbs = batch * prm.N_enc * 16 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        ntt_roofline = {"bound": "hbm", "kernel": "forward NTT, %d transforms of %d points per launch" % (batch, prm.N_enc),
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        del polys

    if rank == 0:
        out = {
            "metric": "prover constraints/sec (ringGroth16, N=8192, 4 RNS primes)",
            "value": round(value, 1), "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "ringGroth16 prover, synthetic chain R1CS m=%d constraints (2^%d per GPU, n_aux=m), ring N=%d L=%d, "
                                   "encodings N_enc=%d K=%d, synthetic CRS %.1f GiB resident in HBM (%.1f GiB per GPU)"
                                   % (m, args.logm, prm.N, prm.L, prm.N_enc, prm.K, (3 * m + 2) * prm.enc_words * 8 / 2**30,
                                      (3 * m + 2) * prm.enc_words * 8 / 2**30 / world),
                       "preset": prm.name, "constraints": m, "parallelism": "limbs%d x terms%d" % (plan.limb_groups, plan.term_shards)},
        }
        if timings:
            out["phase_ms"] = {"witness_map": round(timings["witness_ms"], 3), "msm": round(timings["msm_ms"], 3)}
        if roofline:
            out["roofline"] = roofline
        if ntt_roofline:
            out["ntt_roofline"] = ntt_roofline
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prm, m)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
