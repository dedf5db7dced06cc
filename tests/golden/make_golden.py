#!/usr/bin/env python3
"""Generates the golden fixtures of this directory (data only: inputs and expected outputs).

  python tests/golden/make_golden.py        # run in the build container, from the repo root

1. ref_r1cs_probe.json    -- produced by oracle/_ref/ref_r1cs_probe, i.e. by the REFERENCE'S OWN
   relations/ headers compiled where they lie under /root/reference (oracle/Makefile target `ref`):
   linear_combination::evaluate (relations/variable.tcc:246-254) and is_satisfied on seeded
   circuits.  Needs /root/reference; everything else does not.
2. reference_tests.json   -- the known-answer material of the reference's own tests restated as
   data: util/interpolation_test.cpp:29-55 (nodes 0..7, coefficients 0..7),
   util/division_test.cpp:28-49 (x_i = 2i+1, q_i = i+1, n = 110) and the toy circuit of
   docs/qrp.sage:44-110.  Expected values are computed here with Python integers, independently of
   oracle/ and of the HIP library.
3. oracle_vectors.json    -- seeded inputs (by recipe) and SHA-256 digests + leading words of the
   CPU oracle's outputs for every §8(a) row at test scale.  Pins the oracle against drift on the
   CPU side and lets the GPU box compare the HIP path with committed data.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import oracle as O  # noqa: E402
from ringsnark_amd import params as P  # noqa: E402
from ringsnark_amd import r1cs as R  # noqa: E402
from tests import helpers as H  # noqa: E402


def digest(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "head": [int(x) for x in a.reshape(-1)[:4]], "tail": [int(x) for x in a.reshape(-1)[-2:]]}


# ---- 1. the reference's own headers ------------------------------------------------------------
def probe_case(kind, m, q, seed, slots=4):
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_r1cs_probe")
    cs = R.wide_r1cs(m, [q], seed=seed) if kind == "wide" else R.chain_r1cs(m, [q])
    rng = np.random.RandomState(seed)
    mats = {}
    for name in "abc":
        rp, col, cf = cs.mats[name]
        mats[name] = {"row_ptr": [int(x) for x in rp], "col": [int(x) for x in col], "coeff": [int(x) for x in cf[0]]}
    case = {"kind": kind, "q": q, "m": m, "n_vars": cs.n_vars, "n_inputs": cs.n_inputs, "mats": mats, "slots": []}
    for s in range(slots):
        vals = [int(rng.randint(1, 2**31)) for _ in range(2)]
        for i in range(m):  # forward-solve so the system is satisfied
            def lc(name):
                rp, col, cf = cs.mats[name]
                return sum(int(cf[0, e]) * (1 if col[e] == 0 else vals[col[e] - 1]) for e in range(rp[i], rp[i + 1])) % q
            vals.append(lc("a") * lc("b") % q)
        if s == slots - 1:
            vals[-1] = (vals[-1] + 1) % q  # last slot: tampered, must be reported unsatisfied
        lines = ["%d %d %d %d" % (q, m, cs.n_vars, cs.n_inputs)]
        for name in "abc":
            rp, col, cf = cs.mats[name]
            for i in range(m):
                terms = ["%d %d" % (col[e], cf[0, e]) for e in range(rp[i], rp[i + 1])]
                lines.append("%d %s" % (len(terms), " ".join(terms)))
        lines.append(" ".join(str(x) for x in vals))
        out = subprocess.run([probe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
        case["slots"].append({"assignment": vals, "satisfied": int(out[0].split()[1]),
                              "rows": [[int(x) for x in out[1 + i].split()] for i in range(m)]})
    return case


def probe_case_poly(m, q, seed, S=8):
    """Coefficients that are general ring elements (relations/variable.tcc:246-254 multiplies by any RingT; the DFT
    constraint of benchmarks/bench_ntt_SEAL.cpp:46-53): the reference's headers instantiated over Z_q^S with
    slot-wise operations (the probe's "vec" form).  The LAST slot of the last variable is tampered."""
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_r1cs_probe")
    cs = R.wide_poly_r1cs(m, [q], S, seed=seed)
    rng = np.random.RandomState(seed)
    coeff = lambda name, e: [int(x) for x in cs.poly_table[cs.poly_idx[name][e], 0]] if cs.poly_idx[name][e] >= 0 else [int(cs.mats[name][2][0, e])] * S
    vals = [[int(rng.randint(1, 2**31)) for _ in range(S)] for _ in range(2)]  # [variable][slot]
    for i in range(m):
        def lc(name, s):
            rp, col, _ = cs.mats[name]
            return sum(coeff(name, e)[s] * (1 if col[e] == 0 else vals[col[e] - 1][s]) for e in range(rp[i], rp[i + 1])) % q
        vals.append([lc("a", s) * lc("b", s) % q for s in range(S)])
    case = {"kind": "wide_poly", "q": q, "S": S, "m": m, "n_vars": cs.n_vars, "n_inputs": cs.n_inputs, "mats": {}, "runs": []}
    for name in "abc":
        rp, col, _ = cs.mats[name]
        case["mats"][name] = {"row_ptr": [int(x) for x in rp], "col": [int(x) for x in col],
                              "coeff": [coeff(name, e) for e in range(len(col))],  # S residues per non-zero
                              "is_poly": [int(cs.poly_idx[name][e] >= 0) for e in range(len(col))]}
    for tamper in (False, True):
        v = [list(x) for x in vals]
        if tamper:
            v[-1][-1] = (v[-1][-1] + 1) % q
        lines = ["vec %d %d %d %d %d" % (q, S, m, cs.n_vars, cs.n_inputs)]
        for name in "abc":
            rp, col, _ = cs.mats[name]
            for i in range(m):
                terms = ["%d %s" % (col[e], " ".join(str(x) for x in coeff(name, e))) for e in range(rp[i], rp[i + 1])]
                lines.append("%d %s" % (len(terms), " ".join(terms)))
        lines.append(" ".join(str(x) for var in v for x in var))
        out = subprocess.run([probe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
        case["runs"].append({"assignment": v, "satisfied": int(out[0].split()[1]),
                             "rows": [[[int(x) for x in tok.split(",")] for tok in out[1 + i].split()] for i in range(m)]})
    return case


# ---- 2. the reference's own test data ----------------------------------------------------------
def horner(coeffs, x, q):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % q
    return acc


def reference_tests():
    q = 0xFFFFEE001
    out = {"q": q}
    coeffs = list(range(8))
    out["interpolation_test"] = {"cite": "ringsnark/util/interpolation_test.cpp:29-55", "nodes": list(range(8)),
                                 "coefficients": coeffs, "values": [horner(coeffs, x, q) for x in range(8)]}
    n = 110
    den = [2 * i + 1 for i in range(n)]
    quo = [i + 1 for i in range(n)]
    prod = [0] * (2 * n - 1)
    for i, a in enumerate(den):
        for j, b in enumerate(quo):
            prod[i + j] = (prod[i + j] + a * b) % q
    out["division_test"] = {"cite": "ringsnark/util/division_test.cpp:28-49", "n": n,
                            "divisor": den, "quotient": quo, "product": prod}
    # docs/qrp.sage:44-67: 6 wires, 2 gates.  v/w/y hold each wire's values at the two nodes
    # (r5, r6 are symbolic there; the reference's domain fixes them to 0, 1,
    # util/evaluation_domain.tcc:8-13): gate 1 (node 0): c3 * c4 = c5; gate 2 (node 1): (c1 + c2) * c5 = c6.
    v = [[0, 1], [0, 1], [1, 0], [0, 0], [0, 0], [0, 0]]
    w = [[0, 0], [0, 0], [0, 0], [1, 0], [0, 1], [0, 0]]
    y = [[0, 0], [0, 0], [0, 0], [0, 0], [1, 0], [0, 1]]
    c = [2, 3, 4, 5, 20, 100]  # a satisfying assignment: 4*5 = 20, (2+3)*20 = 100

    def combine(mat):  # sum_i c_i * interpolant of wire i through nodes (0,1): f = f0 + (f1 - f0) x
        f0 = sum(ci * r[0] for ci, r in zip(c, mat)) % q
        f1 = sum(ci * r[1] for ci, r in zip(c, mat)) % q
        return [f0, (f1 - f0) % q]
    V, W, Y = combine(v), combine(w), combine(y)
    num = [(V[0] * W[0] - Y[0]) % q, (V[0] * W[1] + V[1] * W[0] - Y[1]) % q, (V[1] * W[1]) % q]
    h = num[2]  # numerator = h * (x^2 - x), degree 2 -> constant h
    assert num[0] == 0 and (num[1] + h) % q == 0
    out["qrp_sage_toy"] = {"cite": "docs/qrp.sage:44-110", "v": v, "w": w, "y": y, "n_inputs": 4, "assignment": c,
                           "V": V, "W": W, "Y": Y, "t": [0, q - 1, 1], "h": [h]}
    return out


# ---- 3. oracle vectors -------------------------------------------------------------------------
def oracle_vectors():
    v = {"recipes": "inputs: Ctx.random_ring(seed[,count]) / Ctx.random_enc(seed[,count]) of oracle/oracle.py "
                    "(numpy RandomState(seed).randint(0, 2**62) % prime); circuits: ringsnark_amd.r1cs.wide_r1cs / chain_r1cs; "
                    "assignments: tests.helpers.make_assignment(ctx, cs, seed=7)"}
    # C1: one forward NTT, N = 4096, q = 0xffffee001, input i -> (i * 0x9E3779B97F4A7C15 mod 2^64) mod q
    q = 0xFFFFEE001
    x = (np.arange(4096, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(q)
    t = O.NTT(12, q)
    v["C1_ntt"] = {"q": q, "logn": 12, "input": digest(x), "forward": digest(t.fwd(x))}
    for name in ("toy", "toy49"):
        prm = P.preset(name)
        ctx = H.oracle_ctx(prm)
        e = {"params": {"N": prm.N, "q": [int(x) for x in prm.q], "N_enc": prm.N_enc, "Q": [int(x) for x in prm.Q]}}
        a, b = ctx.random_ring(11, 5), ctx.random_ring(12, 5)
        e["ring_mul(11,12;5)"] = digest(ctx.ring_mul(a, b))
        e["ring_sub(11,12;5)"] = digest(ctx.ring_sub(a, b))
        e["batch_encode(limb0, ring11[0])"] = digest(ctx.batch_encode(0, a[0, 0]))
        encs, rings = ctx.random_enc(21, 6), ctx.random_ring(22, 6)
        rings[2] = 0  # an is_zero term (seal_ring.tcc:391-396)
        ip, used = ctx.inner_product(encs, rings)
        e["inner_product(enc21, ring22 with term 2 zero; 6)"] = dict(digest(ip), used=used)
        m = 12
        cs = R.wide_r1cs(m, prm.q)
        asg = H.make_assignment(ctx, cs)
        ds = [ctx.random_ring(60 + k) for k in range(3)]
        wm = {}
        for limb in range(prm.L):
            w = O.witness_map(prm.q[limb], H.oracle_cs(cs), limb, np.ascontiguousarray(asg[:, limb, :]),
                              *[np.ascontiguousarray(d[limb]) for d in ds])
            for k in ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H"):
                wm.setdefault(k, []).append(digest(w[k]))
        e["witness_map(wide m=12, d=60..62)"] = wm
        pk = dict(s_pows=ctx.random_enc(71, m + 1), delta_ts=ctx.random_enc(72, m + 1), delta_mid=ctx.random_enc(73, cs.n_aux),
                  alpha=ctx.random_enc(74), beta=ctx.random_enc(75))
        proof, empty = O.groth16_prove(ctx, H.oracle_cs(cs), pk, asg)
        e["groth16_prove(wide m=12, pk 71..75)"] = {"proof": digest(proof), "empty": [int(x) for x in empty]}
        pk = dict(s_pows=ctx.random_enc(81, m + 1), alpha_s_pows=ctx.random_enc(82, m + 1), beta_prods=ctx.random_enc(83, cs.n_aux),
                  beta_rv_ts=ctx.random_enc(84), beta_rw_ts=ctx.random_enc(85), beta_ry_ts=ctx.random_enc(86))
        proof, empty = O.rinocchio_prove(ctx, H.oracle_cs(cs), pk, asg, *ds)
        e["rinocchio_prove(wide m=12, pk 81..86, d=60..62)"] = {"proof": digest(proof), "empty": [int(x) for x in empty]}
        v[name] = e
    return v


def main():
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_r1cs_probe")):
        cases = [probe_case("wide", 9, 0xFFFFEE001, 5), probe_case("chain", 6, 0xFFFFEE001, 6), probe_case("wide", 17, 0xFFFFC4001, 8),
                 probe_case_poly(7, 0xFFFFEE001, 21), probe_case_poly(12, 0xFFFFC4001, 22, S=4)]
        json.dump({"generator": "oracle/_ref/ref_r1cs_probe (reference relations/ headers, compiled as they lie)", "cases": cases},
                  open(os.path.join(HERE, "ref_r1cs_probe.json"), "w"), indent=0)
    else:
        print("oracle/_ref/ref_r1cs_probe missing: ref_r1cs_probe.json left untouched")
    json.dump(reference_tests(), open(os.path.join(HERE, "reference_tests.json"), "w"), indent=0)
    json.dump(oracle_vectors(), open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
