"""The oracle against the reference's known-answer material and algebraic identities (CPU)."""
import numpy as np
import pytest

from oracle import oracle as O
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H


def test_prime_search_reproduces_reference_primes():
    # docs/qrp.sage:3-5 writes down BFVDefault(4096): the only primes the reference spells out.
    assert O.get_primes(8192, 36, 2) == [0xFFFFEE001, 0xFFFFC4001]
    assert O.get_primes(8192, 37, 1) == [0x1FFFFE0001]
    assert P.get_primes(8192, 36, 2) == [0xFFFFEE001, 0xFFFFC4001]
    for name in ("C2", "C3", "toy", "toy49"):
        prm = P.preset(name)
        assert O.coeff_modulus_create(2 * prm.N_enc, [p.bit_length() for p in prm.Q]) is not None
        prm.validate()


def _br(x, b):
    return int(format(x, "0%db" % b)[::-1], 2)


@pytest.mark.parametrize("logn,bits", [(4, 20), (6, 36), (8, 49)])
def test_ntt_is_evaluation_at_odd_powers_in_bitreversed_order(logn, bits):
    n = 1 << logn
    q = O.get_primes(2 * n, bits, 1)[0]
    t = O.NTT(logn, q)
    psi = O.minimal_primitive_root(2 * n, q)
    assert pow(psi, n, q) == q - 1
    a = np.array([(i * 0x9E3779B97F4A7C15) % q for i in range(n)], dtype=np.uint64)
    A = t.fwd(a)
    for j in range(n):
        e = 2 * _br(j, logn) + 1
        assert int(A[j]) == sum(int(a[k]) * pow(psi, e * k, q) for k in range(n)) % q
    assert (t.inv(A) == a).all()


def test_ntt_negacyclic_convolution():
    logn, n = 6, 64
    q = O.get_primes(2 * n, 30, 1)[0]
    t = O.NTT(logn, q)
    rng = np.random.RandomState(0)
    a = rng.randint(0, q, n).astype(np.uint64)
    b = rng.randint(0, q, n).astype(np.uint64)
    prod = np.array([int(x) * int(y) % q for x, y in zip(t.fwd(a), t.fwd(b))], dtype=np.uint64)
    got = t.inv(prod)
    ref = [0] * n
    for i in range(n):
        for j in range(n):
            k, s = (i + j, 1) if i + j < n else (i + j - n, -1)
            ref[k] = (ref[k] + s * int(a[i]) * int(b[j])) % q
    assert [int(x) for x in got] == ref


def test_interpolation_known_answer():
    # util/interpolation_test.cpp:29-55: nodes 0..7, coefficients 0..7.
    q = 0xFFFFEE001
    coeffs = np.arange(8, dtype=np.uint64).reshape(8, 1)
    y = np.stack([O.poly_eval(q, coeffs, x) for x in range(8)])
    got = O.interpolate(q, y)
    assert (got == coeffs).all()
    for x in range(8):
        assert (O.poly_eval(q, got, x) == y[x]).all()


def test_lagrange_known_answer():
    # util/interpolation_test.cpp:57-83: sum_j y_j L_j(s) == eval(coeffs, s), s = m .. m+19.
    q, n = 0xFFFFEE001, 8
    coeffs = np.arange(n, dtype=np.uint64).reshape(n, 1)
    y = [int(O.poly_eval(q, coeffs, x)[0]) for x in range(n)]
    for s in range(n, n + 20):
        acc = 0
        for j in range(n):
            num = den = 1
            for i in range(n):
                if i != j:
                    num = num * (s - i) % q
                    den = den * (j - i) % q
            acc = (acc + y[j] * num * pow(den, q - 2, q)) % q
        assert acc == int(O.poly_eval(q, coeffs, s)[0])


def test_division_known_answer():
    # util/division_test.cpp:28-49: n = 110, x_i = 2i+1, quotient_i = i+1.
    q, n = 0xFFFFEE001, 110
    x = np.array([2 * i + 1 for i in range(n)], dtype=np.uint64).reshape(n, 1)
    quo = np.array([i + 1 for i in range(n)], dtype=np.uint64).reshape(n, 1)
    y = O.poly_mul(q, quo, x)
    got, nq = O.poly_div_general(q, y, x)
    assert nq <= n
    assert (got[:n] == quo).all()


def test_qrp_sage_toy_circuit():
    # docs/qrp.sage:44-110: 6 wires, 2 gates over GF(0xffffee001); r5, r6 are symbolic there,
    # the reference's domain fixes them to 0, 1 (evaluation_domain.tcc:8-13).
    # gate 1 (node 0): c3 * c4 = c5 ; gate 2 (node 1): (c1 + c2) * c5 = c6.
    q = 0xFFFFEE001
    rows = {"a": [[(3, 1)], [(1, 1), (2, 1)]], "b": [[(4, 1)], [(5, 1)]], "c": [[(5, 1)], [(6, 1)]]}
    cs = R.from_rows(2, 6, 4, rows, [q])
    c = [2, 3, 4, 5, 20, 100]
    asg = np.array(c, dtype=np.uint64).reshape(6, 1)
    w = O.witness_map(q, H.oracle_cs(cs), 0, asg)
    V = [(int(w["A_io"][k, 0]) + int(w["A_mid"][k, 0])) % q for k in range(2)]
    W = [(int(w["B_io"][k, 0]) + int(w["B_mid"][k, 0])) % q for k in range(2)]
    Y = [(int(w["C_io"][k, 0]) + int(w["C_mid"][k, 0])) % q for k in range(2)]
    # V(x) interpolates (c3, c1+c2) on nodes (0,1), etc.
    assert V == [4, (5 - 4) % q] and W == [5, (20 - 5) % q] and Y == [20, (100 - 20) % q]
    assert [int(z) for z in w["Z"]] == [0, q - 1, 1]  # t = x(x-1)
    # h = (V*W - Y) / t : degree 0 here -> h0 = V1*W1
    assert int(w["H"][0, 0]) == V[1] * W[1] % q and int(w["H"][1, 0]) == 0 and int(w["H"][2, 0]) == 0


@pytest.mark.parametrize("m,zk", [(1, False), (5, False), (12, True), (16, True)])
def test_witness_map_identities(m, zk):
    prm = P.preset("toy")
    ctx = H.oracle_ctx(prm)
    cs = R.wide_r1cs(m, prm.q) if m > 1 else R.chain_r1cs(m, prm.q)
    asg = H.make_assignment(ctx, cs)
    for limb in range(prm.L):
        q = prm.q[limb]
        a = np.ascontiguousarray(asg[:, limb, :])
        ds = [np.ascontiguousarray(ctx.random_ring(40 + k)[limb]) for k in range(3)] if zk else [None] * 3
        w = O.witness_map(q, H.oracle_cs(cs), limb, a, *ds)
        ocs = H.oracle_cs(cs)
        full = {}
        for k, nm in enumerate("ABC"):
            ev = O.r1cs_evaluate(q, ocs, k, limb, a)
            full[nm] = O.interpolate(q, ev)
            # linearity of interpolate; index-0 (constant) terms are evaluated in BOTH the io and the
            # mid pass of the reference (r1cs_to_qrp.tcc:175-179,197-201), so they count twice there
            const = O.interpolate(q, O.r1cs_evaluate(q, ocs, k, limb, np.zeros_like(a)))
            tot = (w[nm + "_io"].astype(object) + w[nm + "_mid"].astype(object)) % q
            assert (tot == (full[nm].astype(object) + const.astype(object)) % q).all()
            for x in range(m):
                assert (O.poly_eval(q, full[nm], x) == ev[x]).all()
        # Z = prod (x - i)
        for x in range(m):
            assert int(O.poly_eval(q, w["Z"].reshape(-1, 1), x)[0]) == 0
        assert int(w["Z"][m]) == 1
        # H*Z == (A + d1 Z)(B + d2 Z) - (C + d3 Z)   (r1cs_to_qrp.tcc:123-131)
        S = prm.N
        Zs = np.repeat(w["Z"].reshape(-1, 1), S, axis=1)
        def shift(P_, d):
            out = np.zeros((m + 1, S), dtype=object)
            out[:m] = P_.astype(object)
            if d is not None:
                out = (out + Zs.astype(object) * d.astype(object)[None, :]) % q
            return (out % q).astype(np.uint64)
        A_, B_, C_ = shift(full["A"], ds[0]), shift(full["B"], ds[1]), shift(full["C"], ds[2])
        lhs = O.poly_mul(q, w["H"], Zs)
        rhs = O.poly_mul(q, A_, B_).astype(object)
        rhs[: m + 1] = (rhs[: m + 1] - C_.astype(object)) % q
        assert (lhs.astype(object)[: 2 * m + 1] == rhs % q).all()


def test_batch_encode_decode_and_homomorphism():
    prm = P.preset("toy")
    ctx = H.oracle_ctx(prm)
    sk = ctx.keygen(1)
    T = 6
    r, s = ctx.random_ring(3, T), ctx.random_ring(4, T)
    r[2] = 0
    encs = ctx.enc_encode(sk, s, 5)
    for t in range(T):  # tests/encoding_test.cpp:28-49 (encode -> decode round trip)
        assert (ctx.enc_decode(sk, encs[t]) == s[t]).all()
    out, used = ctx.inner_product(encs, r)
    assert used == T - 1
    acc = np.zeros(ctx.ring_shape(), dtype=np.uint64)
    for t in range(T):
        acc = ctx.ring_add(acc, ctx.ring_mul(s[t], r[t]))
    assert (ctx.enc_decode(sk, out) == acc).all()
    # all-zero coefficients -> EMPTY element (seal_ring.tcc:412,432)
    out0, used0 = ctx.inner_product(encs, np.zeros_like(r))
    assert used0 == 0 and not out0.any()
    # Scalar-1 fast path leaves the ciphertext untouched (seal_ring.tcc:525-527)
    kinds = np.zeros(T, dtype=np.uint8)
    kinds[1] = O.KIND_ONE
    out1, _ = ctx.inner_product(encs[1:2], r[1:2], kinds[1:2])
    assert (out1 == encs[1]).all()


def test_groth16_prover_decodes_to_ring_inner_products():
    prm = P.preset("toy")
    ctx = H.oracle_ctx(prm)
    m = 6
    cs = R.wide_r1cs(m, prm.q)
    asg = H.make_assignment(ctx, cs)
    sk = ctx.keygen(9)
    s_pows_r, dts_r, dmid_r = ctx.random_ring(21, m + 1), ctx.random_ring(22, m + 1), ctx.random_ring(23, cs.n_aux)
    al_r, be_r = ctx.random_ring(24), ctx.random_ring(25)
    pk = dict(s_pows=ctx.enc_encode(sk, s_pows_r, 31), delta_ts=ctx.enc_encode(sk, dts_r, 32),
              delta_mid=ctx.enc_encode(sk, dmid_r, 33), alpha=ctx.enc_encode(sk, al_r, 34)[0],
              beta=ctx.enc_encode(sk, be_r, 35)[0])
    proof, empty = O.groth16_prove(ctx, H.oracle_cs(cs), pk, asg)
    assert empty == [0, 0, 0]
    ocs = H.oracle_cs(cs)
    exp = [al_r.copy(), be_r.copy(), np.zeros(ctx.ring_shape(), dtype=np.uint64)]
    for limb in range(prm.L):
        w = O.witness_map(prm.q[limb], ocs, limb, np.ascontiguousarray(asg[:, limb, :]))
        for t in range(m + 1):
            for which, (vec_io, vec_mid, crs) in enumerate(((w["A_io"], w["A_mid"], s_pows_r), (w["B_io"], w["B_mid"], s_pows_r))):
                if t < m:
                    co = (vec_io[t].astype(object) + vec_mid[t].astype(object)) % prm.q[limb]
                    exp[which][limb] = ((exp[which][limb].astype(object) + crs[t, limb].astype(object) * co) % prm.q[limb]).astype(np.uint64)
            exp[2][limb] = ((exp[2][limb].astype(object) + dts_r[t, limb].astype(object) * w["H"][t].astype(object)) % prm.q[limb]).astype(np.uint64)
        for t in range(cs.n_aux):
            exp[2][limb] = ((exp[2][limb].astype(object) + dmid_r[t, limb].astype(object) * asg[cs.n_inputs + t, limb].astype(object)) % prm.q[limb]).astype(np.uint64)
    for k in range(3):
        assert (ctx.enc_decode(sk, proof[k]) == exp[k]).all(), k


def test_r1cs_evaluate_matches_reference_headers():
    """oracle/_ref/ref_r1cs_probe is built from the reference's own relations/ headers (as they lie
    under /root/reference); it pins linear_combination::evaluate and is_satisfied."""
    import os
    import subprocess
    probe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_r1cs_probe")
    if not os.path.exists(probe):
        pytest.skip("oracle/_ref not built (reference sources absent)")
    q = 0xFFFFEE001
    m = 9
    cs = R.wide_r1cs(m, [q])
    rng = np.random.RandomState(5)
    vals = [int(rng.randint(1, 2**31)) for _ in range(2)]
    # forward-solve over Z_q so the system is satisfied
    for i in range(m):
        rp, col, cf = cs.mats["a"]
        a = sum(int(cf[0, e]) * (1 if col[e] == 0 else vals[col[e] - 1]) for e in range(rp[i], rp[i + 1])) % q
        vals.append(a * vals[i + 1] % q)
    for tamper in (False, True):
        v = list(vals)
        if tamper:
            v[-1] = (v[-1] + 1) % q
        lines = ["%d %d %d %d" % (q, m, cs.n_vars, cs.n_inputs)]
        for name in "abc":
            rp, col, cf = cs.mats[name]
            for i in range(m):
                terms = ["%d %d" % (col[e], cf[0, e]) for e in range(rp[i], rp[i + 1])]
                lines.append("%d %s" % (len(terms), " ".join(terms)))
        lines.append(" ".join(str(x) for x in v))
        out = subprocess.run([probe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
        assert out[0] == "sat %d" % (0 if tamper else 1)
        asg = np.array(v, dtype=np.uint64).reshape(-1, 1)
        ocs = H.oracle_cs(cs)
        ev = [O.r1cs_evaluate(q, ocs, k, 0, asg) for k in range(3)]
        for i in range(m):
            assert [int(x) for x in out[1 + i].split()] == [int(ev[k][i, 0]) for k in range(3)]


@pytest.mark.parametrize("name,T", [("toy", 21), ("toy49", 12), ("toy60", 9), ("C2", 3)])
def test_seal_style_timing_library_equals_the_checker(name, T):
    """oracle/librs_oracle_fast.so -- the TIMED leg of bench.py's cpu_baseline (Harvey lazy NTT with Shoup quotients,
    Barrett products: the arithmetic SEAL publishes) -- must give the checker's results bit for bit: transforms, the
    inner product (zero term, Scalar-1 term, tiled window, threads) and the O(m^2) witness map."""
    from oracle import fastcpu as F
    prm = P.preset(name)
    ctx = H.oracle_ctx(prm)
    f = F.FastCtx(prm.N, prm.q, prm.N_enc, prm.Q)
    logn = prm.N_enc.bit_length() - 1
    for modset, primes in ((0, prm.q), (1, prm.Q)):
        for idx, p in enumerate(primes):
            x = np.random.RandomState(idx).randint(0, p, prm.N_enc, dtype=np.int64).astype(np.uint64)
            t = O.NTT(logn, p)
            assert (f.ntt(modset, idx, x) == t.fwd(x.copy())).all() and (f.ntt(modset, idx, x, True) == t.inv(x.copy())).all()
    encs, rings = ctx.random_enc(31, T), ctx.random_ring(32, T)
    kinds = np.zeros(T, dtype=np.uint8)
    rings[1] = 0
    kinds[2] = O.KIND_ONE
    exp, used = ctx.inner_product(encs, rings, kinds)
    for threads in (1, 3):
        got, u = f.inner_product(encs, rings, kinds, threads=threads)
        assert u == used and (got == exp).all()
    exp, used = ctx.inner_product(encs[:2], rings, kinds, threads=0, window=2)
    got, u = f.inner_product(encs[:2], rings, kinds, threads=0, window=2)
    assert u == used and (got == exp).all()
    # one ring limb, all its 2 K slabs in one pass (the complete MSM check of a headline proof, tests/proof_check.py): equal to the
    # checker's whole inner product on that limb and to its per-slab form, in two term ranges over a tiled key
    rows_all = ctx.random_ring(33, T)
    rows_all[min(3, T - 1)] = 0
    W = min(5, T)
    exp, _ = ctx.inner_product(encs[:W], rows_all, None, threads=0, window=W)
    for limb in range(prm.L):
        key = np.ascontiguousarray(encs[:W, limb])
        acc = np.zeros((2, prm.K, prm.N_enc), dtype=np.uint64)
        cut = T // 2
        f.inner_product_limb(limb, key, rows_all[:cut, limb], acc, t0=0, window=W, threads=2)
        f.inner_product_limb(limb, key, rows_all[cut:, limb], acc, t0=cut, window=W, threads=0)
        assert (acc == exp[limb]).all()
        slab = np.zeros(prm.N_enc, dtype=np.uint64)
        ctx.inner_product_slab(limb, prm.K - 1, np.ascontiguousarray(key[:, 1, prm.K - 1]), rows_all[:, limb], slab, window=W)
        assert (slab == acc[1, prm.K - 1]).all()
    if prm.N <= 64:
        cs = R.wide_r1cs(25, prm.q)
        asg = H.make_assignment(ctx, cs)
        ds = [ctx.random_ring(60 + k) for k in range(3)]
        limb = prm.L - 1
        args = (prm.q[limb], H.oracle_cs(cs), limb, np.ascontiguousarray(asg[:, limb, :])) + tuple(np.ascontiguousarray(d[limb]) for d in ds)
        w0, w1 = O.witness_map(*args), F.witness_map(*args, threads=2)
        assert all((w0[k] == w1[k]).all() for k in w0)
