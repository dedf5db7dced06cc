#!/usr/bin/env python3
"""Golden vectors at CONFIGURATION scale (BASELINE.json configs[1]: ringGroth16 prover, 2^10 constraints, ring
N = 4096 / 2 primes, encodings N_enc = 8192 / K = 4 -- preset C2).

  python tests/golden/make_config_golden.py      # build container, repo root; a few minutes on 8 cores

Inputs are BY RECIPE (nothing large is stored): the synthetic chain R1CS of SURVEY.md 8(d), an assignment and a
proving key filled by the generators of the benchmark harness (rs_fill_uniform / rs_chain_assignment), which
this script restates in numpy so that the CPU oracle and the device start from identical bytes.  Outputs: SHA-256
digests (+ leading words) of the oracle's witness-map vectors and of the proof {A, B, C}.  The proof is composed
from the oracle's OpenMP forms exactly as rso_groth16_prove composes it (groth16.tcc:89-112); the script first
asserts, at toy scale, that this composition equals rso_groth16_prove bit for bit.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import oracle as O  # noqa: E402
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from tests import helpers as H  # noqa: E402

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    """splitmix64 finaliser, as ringsnark_amd/csrc/prover.hip mix64 (uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fill_uniform(shape, inner, mods, seed):
    """rs_fill_uniform: word i of the array = mix64(mix64(seed) ^ i) mod mods[(i / inner) % len(mods)]."""
    words = int(np.prod(shape))
    i = np.arange(words, dtype=np.uint64)
    p = np.array(mods, dtype=np.uint64)[(i // np.uint64(inner)) % np.uint64(len(mods))]
    s = mix64(np.array([seed], dtype=np.uint64))[0]
    return (mix64(s ^ i) % p).reshape(shape)


def digest(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "head": [int(x) for x in a.reshape(-1)[:4]], "tail": [int(x) for x in a.reshape(-1)[-2:]]}


SEEDS = {"x": 7, "s_pows": 13, "delta_ts": 14, "delta_mid": 15, "alpha": 16, "beta": 17}


def inputs(prm, m, ctx):
    cs = R.chain_r1cs(m, prm.q)
    asg = np.empty((m + 2, prm.L, prm.N), dtype=np.uint64)
    asg[:2] = fill_uniform((2, prm.L, prm.N), prm.N, prm.q, SEEDS["x"])
    for i in range(m):  # rs_chain_assignment: x_{i+2} = x_i * x_{i+1}
        asg[i + 2] = ctx.ring_mul(asg[i], asg[i + 1])
    enc = lambda count, seed: fill_uniform((count, prm.L, 2, prm.K, prm.N_enc), prm.N_enc, prm.Q, seed)
    pk = dict(s_pows=enc(m + 1, SEEDS["s_pows"]), delta_ts=enc(m + 1, SEEDS["delta_ts"]), delta_mid=enc(cs.n_aux, SEEDS["delta_mid"]),
              alpha=enc(1, SEEDS["alpha"])[0], beta=enc(1, SEEDS["beta"])[0])
    return cs, asg, pk


def prove_mt(ctx, prm, cs, asg, pk):
    """groth16.tcc:70-115 from the oracle's OpenMP forms."""
    m, ocs = cs.m, H.oracle_cs(cs)
    w = {k: np.zeros((m + (k == "H"), prm.L, prm.N), dtype=np.uint64) for k in ("A_io", "A_mid", "B_io", "B_mid", "C_io", "C_mid", "H")}
    for limb in range(prm.L):
        o = O.witness_map(prm.q[limb], ocs, limb, np.ascontiguousarray(asg[:, limb, :]), threads=0)
        for k in w:
            w[k][:, limb, :] = o[k]

    def ip(key, vec):
        return ctx.inner_product(np.ascontiguousarray(key[:vec.shape[0]]), np.ascontiguousarray(vec), threads=0)

    a = ctx.enc_add(ctx.enc_add(ip(pk["s_pows"], w["A_io"])[0], ip(pk["s_pows"], w["A_mid"])[0]), pk["alpha"])  # :89-95
    b = ctx.enc_add(ctx.enc_add(ip(pk["s_pows"], w["B_io"])[0], ip(pk["s_pows"], w["B_mid"])[0]), pk["beta"])   # :97-103
    c = ip(pk["delta_ts"], w["H"])[0]                                                                           # :105-107
    if cs.n_aux:
        c = ctx.enc_add(c, ip(pk["delta_mid"], asg[cs.n_inputs:])[0])                                           # :108-112
    return np.stack([a, b, c]), w


def main():
    # the composition above == rso_groth16_prove (every inner product non-empty for these inputs)
    prm = P.preset("toy")
    ctx = H.oracle_ctx(prm)
    cs, asg, pk = inputs(prm, 12, ctx)
    exp, _ = O.groth16_prove(ctx, H.oracle_cs(cs), pk, asg)
    got, _ = prove_mt(ctx, prm, cs, asg, pk)
    assert (got == exp).all(), "composition differs from rso_groth16_prove"
    out = {"recipe": "chain R1CS; assignment rows 0,1 = rs_fill_uniform(layout ring, seed %d), rest rs_chain_assignment; key vectors "
                     "rs_fill_uniform(layout enc, seeds %s)" % (SEEDS["x"], {k: v for k, v in SEEDS.items() if k != "x"}),
           "seeds": SEEDS, "cases": {}}
    for name, m in (("toy", 12), ("C2", 1 << 10)):
        prm = P.preset(name)
        ctx = H.oracle_ctx(prm)
        t0 = time.time()
        cs, asg, pk = inputs(prm, m, ctx)
        proof, w = prove_mt(ctx, prm, cs, asg, pk)
        case = {"preset": name, "m": m, "assignment": digest(asg), "s_pows": digest(pk["s_pows"]),
                "witness": {k: digest(v) for k, v in w.items()}, "proof": digest(proof),
                "proof_elements": [digest(proof[k]) for k in range(3)]}
        out["cases"]["%s_m%d" % (name, m)] = case
        print(name, m, "%.0f s" % (time.time() - t0), flush=True)
    json.dump(out, open(os.path.join(HERE, "config_vectors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
